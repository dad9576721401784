// stft.hip.h - framed real STFT / inverse STFT kernels (hand-written HIP for gfx950).
//
// Replaces speechbrain STFT/ISTFT -> torch.stft/torch.istft as called by TemporalCRN.stft_trans /
// istft_trans (reference CRN.py:505-520; semantics restated in SURVEY.md Appendix A2/A3):
//   n_fft-point real FFT, periodic Hamming window of `win` samples centred in the n_fft frame, hop, centre
//   padding n_fft/2 zeros both sides, onesided; inverse = irfft * window, overlap-add, / sum(w^2), trim.
//
// One workgroup owns one K-sample segment of one microphone: the segment is loaded once (coalesced)
// into LDS, each of the T real frames becomes one N/2-point complex transform (even/odd packing) run
// through LDS-resident Stockham radix-5/4/2 passes (fft_lds.h); the onesided spectrum is recovered on the
// way out.  HBM traffic = K*4 B in + T*F*8 B out per segment; everything else stays in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include "fft_lds.h"

namespace se {

constexpr int kFftBatch = 11;  // real frames (= N/2-point complex transforms) resident in LDS per round
constexpr int kMaxRadices = 12;

struct FftPlan {
    int N;      // n_fft (real length); the complex transforms are N/2 long
    int npass;  // radix plan of N/2
    int radices[kMaxRadices];
};

// Runs all passes of `nfft` N/2-point transforms; data starts in a, returns the buffer holding the result.
// tw is the n_fft-point table exp(-2 pi i m / N), used with stride 2.
__device__ inline cf2 *fft_run(cf2 *a, cf2 *b, const FftPlan &pl, int nfft, const cf2 *tw) {
    int Ns = 1;
    const int tid = threadIdx.x, nth = blockDim.x, N2 = pl.N / 2;
    for (int p = 0; p < pl.npass; p++) {
        const int R = pl.radices[p];
        if (R == 5) fft_pass<5>(a, b, N2, N2, nfft, Ns, tw, tid, nth, 2);
        else if (R == 4) fft_pass<4>(a, b, N2, N2, nfft, Ns, tw, tid, nth, 2);
        else fft_pass<2>(a, b, N2, N2, nfft, Ns, tw, tid, nth, 2);
        Ns *= R;
        __syncthreads();
        cf2 *t = a; a = b; b = t;
    }
    return a;
}

struct StftArgs {
    const float *src;       // waveform base; row r = (r / M) * strideB + (r % M) * strideM
    long strideB, strideM;
    int M;
    long off;               // sample k of the segment reads src[row_base + off + k] ...
    long L;                 // ... if 0 <= off + k < L, else 0 (utility.padding zeros, utility.py:312-336)
    int K, T, F, hop;
    cf2 *spec;              // out: element (row, t, f) at spec[row*sR + t*sT + f*sF]
    long sR, sT, sF;
    const float *window;    // [N] hamming(win) centred in N
    const cf2 *tw;          // [N] exp(-2 pi i m / N)
    FftPlan plan;
    long seg_off, seg_spec;  // blockIdx.y = segment of a batch of segments: off += y*seg_off, spec += y*seg_spec
    const long *Lrow;        // optional per-stream lengths [rows / M] (ragged batches): samples beyond a stream's own length read as zeros
};

// LDS: sig[K+N] | win[N] | tw[N] (cf2) | bufA[kFftBatch*N/2] | bufB[kFftBatch*N/2]
inline size_t stft_lds_bytes(int K, int N) {
    return sizeof(float) * (size_t)(K + N + N) + sizeof(cf2) * (size_t)(N + kFftBatch * N);
}

#ifdef SE_AUX_KERNELS  // kernel bodies live in se_aux.hip (compiled without SLP vectorisation, see its header)
__global__ __launch_bounds__(256) void k_stft(StftArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = a.plan.N, N2 = N / 2, K = a.K, T = a.T, F = a.F, pad = N / 2;
    float *sig = reinterpret_cast<float *>(smem);
    float *win = sig + K + N;
    cf2 *tw = reinterpret_cast<cf2 *>(win + N);
    cf2 *bufA = tw + N;
    cf2 *bufB = bufA + kFftBatch * N2;
    const int row = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    const float *src = a.src + (long)(row / a.M) * a.strideB + (long)(row % a.M) * a.strideM;
    const long seg_first = a.off + (long)blockIdx.y * a.seg_off;
    cf2 *spec_out = a.spec + (long)blockIdx.y * a.seg_spec;
    const long Lr = a.Lrow ? min(a.Lrow[row / a.M], a.L) : a.L;
    for (int i = tid; i < K + N; i += nth) {
        const long k = (long)i - pad + seg_first;
        sig[i] = (i >= pad && i < pad + K && k >= 0 && k < Lr) ? src[k] : 0.0f;
    }
    for (int i = tid; i < N; i += nth) { win[i] = a.window[i]; tw[i] = a.tw[i]; }
    __syncthreads();
    for (int t0 = 0; t0 < T; t0 += kFftBatch) {
        const int nf = min(kFftBatch, T - t0);
        // z[n] = x[2n] + i x[2n+1] of the windowed frame: one transform per frame (see fft_lds.h)
        for (int i = tid; i < nf * N2; i += nth) {
            const int f = i / N2, n = i - f * N2;
            const float *s = sig + (t0 + f) * a.hop + 2 * n;
            bufA[i] = cf2{win[2 * n] * s[0], win[2 * n + 1] * s[1]};
        }
        __syncthreads();
        const cf2 *Z = fft_run(bufA, bufB, a.plan, nf, tw);
        for (int i = tid; i < nf * F; i += nth) {
            const int f = i / F, k = i - f * F;
            spec_out[(long)row * a.sR + (long)(t0 + f) * a.sT + (long)k * a.sF] = rfft_post(Z + f * N2, k, N2, tw);
        }
        __syncthreads();
    }
}

#endif  // SE_AUX_KERNELS

struct IstftArgs {
    const cf2 *spec;        // element (row, t, f) at spec[row*sR + t*sT + f*sF]
    long sR, sT, sF;
    int K, T, F, hop;
    float *wav;             // row r at wav + r*wav_ld
    long wav_ld;
    const float *window;    // [N]
    const float *env;       // [K] sum_t w^2 at output sample i (already offset by n_fft/2)
    const cf2 *tw;
    FftPlan plan;
    long seg_spec, seg_wav;  // blockIdx.y = segment of a batch of segments
};

// LDS: frames[T*N] | win[N] | tw[N] (cf2) | bufA | bufB
inline size_t istft_lds_bytes(int T, int N) {
    return sizeof(float) * (size_t)(T * N + N) + sizeof(cf2) * (size_t)(N + kFftBatch * N);
}

#ifdef SE_AUX_KERNELS
__global__ __launch_bounds__(256) void k_istft(IstftArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = a.plan.N, N2 = N / 2, K = a.K, T = a.T, pad = N / 2;
    float *frames = reinterpret_cast<float *>(smem);
    float *win = frames + T * N;
    cf2 *tw = reinterpret_cast<cf2 *>(win + N);
    cf2 *bufA = tw + N;
    cf2 *bufB = bufA + kFftBatch * N2;
    const int row = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    for (int i = tid; i < N; i += nth) { win[i] = a.window[i]; tw[i] = a.tw[i]; }
    __syncthreads();
    const float invN2 = 1.0f / (float)N2;
    for (int t0 = 0; t0 < T; t0 += kFftBatch) {
        const int nf = min(kFftBatch, T - t0);
        for (int i = tid; i < nf * N2; i += nth) {
            const int f = i / N2, k = i - f * N2;
            const cf2 *s = a.spec + (long)blockIdx.y * a.seg_spec + (long)row * a.sR + (long)(t0 + f) * a.sT;
            cf2 xk = s[(long)k * a.sF], xn = s[(long)(N2 - k) * a.sF];
            if (k == 0) { xk.y = 0; xn.y = 0; }  // C2R ignores Im of DC / Nyquist
            bufA[i] = irfft_pre(xk, xn, k, tw);
        }
        __syncthreads();
        const cf2 *R = fft_run(bufA, bufB, a.plan, nf, tw);
        for (int i = tid; i < nf * N2; i += nth) {
            const int f = i / N2, n = i - f * N2;
            const cf2 r = R[i];
            float *fr = frames + (t0 + f) * N + 2 * n;
            fr[0] = (r.x * invN2) * win[2 * n];
            fr[1] = (-r.y * invN2) * win[2 * n + 1];
        }
        __syncthreads();
    }
    for (int i = tid; i < K; i += nth) {
        const int pos = pad + i;
        int t0 = (pos - N + a.hop) / a.hop;  // ceil((pos-N+1)/hop)
        if (t0 < 0) t0 = 0;
        int t1 = pos / a.hop;
        if (t1 > T - 1) t1 = T - 1;
        float s = 0.0f;
        for (int t = t0; t <= t1; t++) s += frames[t * N + (pos - t * a.hop)];
        a.wav[(long)blockIdx.y * a.seg_wav + (long)row * a.wav_ld + i] = s / a.env[i];
    }
}

// utility.over_add on the engine's segment outputs (utility.py:373-403) + the K/2 strip of
// realtime_process (CRN.py:587-588).  yseg [B, Nseg, K] -> out [B, L].
__global__ void k_overlap_avg(const float *yseg, float *out, int Nseg, int K, long L, long skip, const long *Lrow) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= L) return;
    if (Lrow && i >= Lrow[b]) { out[(long)b * L + i] = 0.0f; return; }  // ragged batch: nothing beyond the stream's own length
    const long P = K / 2;
    const long i2 = i + skip, i1 = i2 + P;
    const float *y = yseg + (long)b * Nseg * K;
    const float v1 = y[(2 * (i1 / K)) * K + i1 % K];
    const float v2 = y[(2 * (i2 / K) + 1) * K + i2 % K];
    out[(long)b * L + i] = (v1 + v2) / 2;
}

#endif  // SE_AUX_KERNELS

// host-side launchers, defined in se_aux.hip
void launch_k_stft(dim3 grid, size_t lds, hipStream_t st, const StftArgs &a);
void launch_k_istft(dim3 grid, size_t lds, hipStream_t st, const IstftArgs &a);
void launch_k_overlap_avg(dim3 grid, hipStream_t st, const float *yseg, float *out, int Nseg, int K, long L, long skip, const long *Lrow = nullptr);
void aux_set_fft_lds(int stft_bytes, int istft_bytes);

}  // namespace se
