// se_stoi.hip - the STOI term of TemporalCRN.compute_loss as hand-written kernels, forward and backward (se_loss_stoi_* C ABI).
//
// Reference: utility.stoi_loss utility.py:821-916 (+ thirdoct 480-518, removeSilentFrames 521-571) on torchaudio's
// Resample(16000, 10000) and Spectrogram(512, 256, 128, power=2); restated in losses._stoi_d as ~150 batched torch ops per
// micro-batch (~1 200 tiny kernels per training step, forward + backward of two micro-batches: 4.7 ms of a 24-ms step even
// when replayed as a captured graph, profiles/r03_timeline_train.txt).  Here the same arithmetic is 7 launches forward and 4 backward:
//
//   k_stoi_resample      16 kHz -> 10 kHz, Kaldi LinearResample as a 5-phase FIR (double accumulation like the restatement)
//   k_stoi_frames        256-sample frames every 128 of the CLEAN signal: energies, the -40 dB rule, kept frames in time order
//   k_stoi_env           per (utterance, spectrogram frame): the silence-removed signal is never materialised - a sample of it is the
//                        overlap-add of <= 2 kept, Hann-weighted frames of the 10 kHz signal, gathered on the fly; reflect padding at the
//                        utterance's OWN end; 256-point window inside a 512-point DFT, evaluated directly (256 x 257 complex MACs per
//                        frame from an LDS twiddle table: 0.5 GFLOP per micro-batch, no FFT plumbing); 15 third-octave band envelopes
//   k_stoi_corr_win      the 30-frame clipped, normalised correlation (double), one thread per (band, window): its value AND its
//                        derivative w.r.t. its 30 predicted-envelope entries (kept for the backward: 1.5 MB per micro-batch)
//   k_stoi_corr_sum      fixed-order sum per utterance (bit-reproducible), the fewer-than-30-frames and the too-short (0.99) branches
//   k_stoi_corr_bwd      every (band, frame) sums the <= 30 windows it belongs to - no atomics (a first form that re-derived those
//                        windows per frame in one workgroup per utterance took 3.2 ms per launch)
//   k_stoi_env_bwd       envelope -> |S|^2 -> windowed DFT adjoint per frame
//   k_stoi_gather_bwd    adjoint of (reflect padding o overlap-add o frame selection): every 10 kHz sample gathers its <= 12 terms
//   k_stoi_resample_bwd  adjoint FIR
//
// Gradients flow to the prediction only (the frame selection is a function of the clean signal).  PARITY: pinned by
// tests/golden/loss_golden.npz through the same tests as the torch restatement (unpinned at the torchaudio boundary, see losses.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstdio>

#include "../../include/se_engine.h"

namespace {

constexpr int kPhases = 5, kStride = 8;  // 10000 / 2000, 16000 / 2000
constexpr int kBands = 15, kBins = 257, kWin = 256, kHop = 128, kSeg = 30;
constexpr double kSmall = 2.220446049250313e-16;  // utility.py:478 smallVal = np.finfo("float").eps
constexpr double kClip = 5.62341325;              // 10 ** (15 / 20), utility.py:874
constexpr float kPi = 3.14159265358979323846f;

struct StoiShape {
    int B;
    long L, Lo;
    int Fu, Tm;
    // workspace offsets in floats
    long t10, p10, n10, nk, order, rank, Ot, Op, Sp, dOp, dfr, dp10, cw, dw, total;
    int Mmax;  // windows per band an utterance can have (Tm - 29, at least 1)
};

__host__ __device__ inline StoiShape stoi_shape(int B, long L) {
    StoiShape s{};
    s.B = B; s.L = L;
    s.Lo = (L * 5 + 7) / 8;
    s.Fu = s.Lo >= kWin ? (int)((s.Lo - kWin) / kHop + 1) : 0;
    s.Tm = s.Fu + 2;
    long o = 0;
    auto take = [&](long n) { const long at = o; o += (n + 3) & ~3L; return at; };
    s.t10 = take((long)B * s.Lo); s.p10 = take((long)B * s.Lo);
    s.n10 = take(B); s.nk = take(B);
    s.order = take((long)B * s.Fu); s.rank = take((long)B * s.Fu);
    s.Ot = take((long)B * s.Tm * kBands); s.Op = take((long)B * s.Tm * kBands);
    s.Sp = take((long)B * s.Tm * kBins * 2);
    s.dOp = take((long)B * s.Tm * kBands);
    s.dfr = take((long)B * s.Tm * kWin);
    s.dp10 = take((long)B * s.Lo);
    s.Mmax = s.Tm - (kSeg - 1) > 1 ? s.Tm - (kSeg - 1) : 1;
    s.cw = take((long)B * kBands * s.Mmax * 2);     // per-window correlation (double)
    s.dw = take((long)B * kBands * s.Mmax * kSeg);  // per-window d corr / d Y (float)
    s.total = o;
    return s;
}

// LinearResample::GetNumOutputSamples with ticks of 1 / 80000 s: 5 ticks per input sample, 8 per output sample
__device__ inline int resample_count(long len) {
    const long interval = len * 5;
    if (interval <= 0) return 0;
    const long last = interval / 8;
    return (int)(last * 8 == interval ? last : last + 1);
}

__global__ __launch_bounds__(256) void k_stoi_resample(const float *x, const int64_t *lens, long L, long Lo, const float *w, const int *first, int W,
                                                       float *y, int *n10) {
    const int b = blockIdx.y;
    const long len = lens[b] < L ? (lens[b] > 0 ? lens[b] : 0) : L;
    const int nout = resample_count(len);
    if (n10 && blockIdx.x == 0 && threadIdx.x == 0) n10[b] = nout;
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    if (n >= Lo) return;
    float v = 0.0f;
    if (n < nout) {
        const int i = (int)(n % kPhases);
        const long k = n / kPhases, start = first[i] + kStride * k;
        const float *xb = x + (long)b * L, *wi = w + (long)i * W;
        double acc = 0.0;
        for (int j = 0; j < W; j++) {
            const long m = start + j;
            if (m >= 0 && m < len) acc += (double)xb[m] * (double)wi[j];
        }
        v = (float)acc;
    }
    y[(long)b * Lo + n] = v;
}

// one workgroup per utterance
__global__ __launch_bounds__(1024) void k_stoi_frames(const float *t10, const int *n10, long Lo, int Fu, const float *hann_sym, int *order, int *rank, int *nk_out) {
    extern __shared__ float energy[];  // [Fu]
    __shared__ float red[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *t = t10 + (long)b * Lo;
    const int n = n10[b];
    // floor divisions (torch.div(..., rounding_mode="floor")); n >= 0
    const int n1 = n / kWin, n2 = (n - kHop) >= 0 ? (n - kHop) / kWin : -(((kHop - n) + kWin - 1) / kWin);
    const int nf = max(n1 + n2, 0);
    for (int f = wave; f < Fu; f += 16) {
        float s = 0.0f;
        for (int r = lane; r < kWin; r += 64) {
            const float wv = hann_sym[r], xv = t[(long)f * kHop + r];
            s += (wv * wv) * (xv * xv);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) energy[f] = 20.0f * log10f(sqrtf(s) / 16.0f + (float)kSmall);
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int f = tid; f < Fu && f < nf; f += 1024) mx = fmaxf(mx, energy[f]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_down(mx, off, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
    for (int i = 1; i < 16; i++) mx = fmaxf(mx, red[i]);
    if (tid == 0) {  // kept frames first, in time order (stable): 233 frames, serial
        int cnt = 0;
        for (int f = 0; f < Fu; f++) {
            const bool keep = f < nf && (energy[f] - mx + 40.0f) > 0.0f;
            rank[(long)b * Fu + f] = keep ? cnt : -1;
            if (keep) order[(long)b * Fu + cnt++] = f;
        }
        for (int k = cnt; k < Fu; k++) order[(long)b * Fu + k] = 0;
        nk_out[b] = cnt;
    }
}

// sample s of the silence-removed, overlap-added signal of utterance b (s < 128 (nk + 1))
__device__ inline float stoi_sample(const float *sig, const int *order, int nk, const float *hann_sym, int s) {
    const int k1 = s >> 7;
    float v = 0.0f;
    if (k1 < nk) { const int r = s - (k1 << 7); v += hann_sym[r] * sig[(long)order[k1] * kHop + r]; }
    if (k1 >= 1 && k1 - 1 < nk) { const int r = s - ((k1 - 1) << 7); v += hann_sym[r] * sig[(long)order[k1 - 1] * kHop + r]; }
    return v;
}

// grid (Tm, B): third-octave envelope of spectrogram frame t; S (one-sided 512-point DFT of the 256-sample Hann frame) saved when Sp != nullptr
__global__ __launch_bounds__(256) void k_stoi_env(const float *sig10, long Lo, const int *order_all, const int *nk_all, int Fu, int Tm, const float *hann_sym,
                                                  const float *hann_per, const int *band_lo, const int *band_hi, float *O, float *Sp) {
    __shared__ float xs[kWin];
    __shared__ float2 tw[512];
    __shared__ float pw[kBins + 3];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int nk = nk_all[b], T = nk + 2, Ls = kHop * (nk + 1);
    float *Ob = O + ((long)b * Tm + t) * kBands;
    if (t >= T) {  // beyond the utterance's frames: zeros (tvalid)
        if (tid < kBands) Ob[tid] = 0.0f;
        if (Sp) { float *sp = Sp + ((long)b * Tm + t) * kBins * 2; for (int i = tid; i < kBins * 2; i += 256) sp[i] = 0.0f; }
        return;
    }
    for (int i = tid; i < 512; i += 256) { float sn, cs; sincosf(2.0f * kPi * (float)i / 512.0f, &sn, &cs); tw[i] = make_float2(cs, sn); }
    {
        int n = t * kHop + tid - kHop;  // window sample tid sits at position 128 + tid of the centred 512 frame
        int m = n < 0 ? -n : n;
        if (m >= Ls) m = 2 * (Ls - 1) - m;
        m = min(max(m, 0), Ls - 1);
        xs[tid] = hann_per[tid] * stoi_sample(sig10 + (long)b * Lo, order_all + (long)b * Fu, nk, hann_sym, m);
    }
    __syncthreads();
    float *sp = Sp ? Sp + ((long)b * Tm + t) * kBins * 2 : nullptr;
    for (int k = tid; k < kBins; k += 256) {
        float re = 0.0f, im = 0.0f;
        const int step = k & 511;
        int idx = (kHop * k) & 511;  // (j + 128) k mod 512 at j = 0
        for (int j = 0; j < kWin; j++) {
            const float2 c = tw[idx];
            const float x = xs[j];
            re += x * c.x;
            im -= x * c.y;
            idx = (idx + step) & 511;
        }
        pw[k] = re * re + im * im;
        if (sp) { sp[2 * k] = re; sp[2 * k + 1] = im; }
    }
    __syncthreads();
    if (tid < kBands) {
        float s = 0.0f;
        for (int k = band_lo[tid]; k < band_hi[tid]; k++) s += pw[k];
        Ob[tid] = sqrtf(s + 1e-14f);
    }
}

// The clipped, normalised correlation of one window of n <= 30 entries (X = clean envelope, Y = predicted envelope, fp32 values, double
// arithmetic like the restatement's .double()): returns corr and writes d corr / d Y_i to dy[i] (fp32).  Five passes over the entries
// kept in registers; nothing but scalars is carried between them (a first form with double arrays for yc / d yc spilled to scratch:
// 138 us per launch).  `cnt` = entries that count for the means (= n here).
__device__ inline double stoi_window(const float (&X)[kSeg], const float (&Y)[kSeg], int n, double cnt, float *dy) {
    double nx2 = 0, ny2 = 0;
#pragma unroll
    for (int i = 0; i < kSeg; i++)
        if (i < n) { nx2 += (double)X[i] * X[i]; ny2 += (double)Y[i] * Y[i]; }
    const double nX = sqrt(nx2), nY = sqrt(ny2), a = nX / (nY + kSmall);
    auto clipped = [&](int i, bool &takeu) {
        const double u = (double)Y[i] * a, v = (double)X[i] + (double)X[i] * kClip;
        takeu = u <= v;
        return takeu ? u : v;
    };
    double mx = 0, my = 0;
#pragma unroll
    for (int i = 0; i < kSeg; i++)
        if (i < n) { bool tu; mx += X[i]; my += clipped(i, tu); }
    mx /= cnt; my /= cnt;
    double xc2 = 0, yc2 = 0, xy = 0;
#pragma unroll
    for (int i = 0; i < kSeg; i++)
        if (i < n) { bool tu; const double xc = X[i] - mx, yy = clipped(i, tu) - my; xc2 += xc * xc; yc2 += yy * yy; xy += xc * yy; }
    const double nxc = sqrt(xc2), nyc = sqrt(yc2);
    const double ix = 1.0 / (nxc + kSmall), iy = 1.0 / (nyc + kSmall);
    const double corr = xy * ix * iy;
    if (dy) {
        // g_i = d corr / d yn_i = xn_i,  yn = ycm / (nyc + eps):  d_i = g_i iy - (sum_j g_j ycm_j) ycm_i / (nyc (nyc + eps)^2), then centring
        const double gdot = xy * ix, k2 = nyc > 0 ? gdot * iy * iy / nyc : 0.0;
        double mean_d = 0, du_dot = 0;
#pragma unroll
        for (int i = 0; i < kSeg; i++)
            if (i < n) { bool tu; const double yy = clipped(i, tu) - my; mean_d += (X[i] - mx) * ix * iy - k2 * yy; }
        mean_d /= cnt;
#pragma unroll
        for (int i = 0; i < kSeg; i++)
            if (i < n) {
                bool tu;
                const double yy = clipped(i, tu) - my;
                const double d = (X[i] - mx) * ix * iy - k2 * yy - mean_d;
                if (tu) du_dot += d * Y[i];  // min(u, v): the gradient goes to u where u is the smaller
            }
        const double da = nY > 0 ? -nX / ((nY + kSmall) * (nY + kSmall)) / nY : 0.0;  // d a / d Y_k = da * Y_k
#pragma unroll
        for (int i = 0; i < kSeg; i++) {
            float out = 0.0f;
            if (i < n) {
                bool tu;
                const double yy = clipped(i, tu) - my;
                const double d = tu ? (X[i] - mx) * ix * iy - k2 * yy - mean_d : 0.0;
                out = (float)(a * d + du_dot * da * Y[i]);
            }
            dy[i] = out;
        }
    }
    return corr;
}

__device__ inline double block_sum_d(double v, double *red, int nwaves) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0;
    for (int i = 0; i < nwaves; i++) s += red[i];
    return s;
}

// window geometry of an utterance: Mi windows per band of n entries each (30-frame windows, or ONE vector over all T < 30 frames)
__device__ inline void stoi_windows(int T, int Tm, int &Mi, int &n) {
    if (T >= kSeg) { Mi = T - (kSeg - 1); n = kSeg; }
    else { Mi = 1; n = min(T, min(Tm, kSeg - 1)); }
}

// grid (ceil(15 Mmax / 256), B): one thread per (band, window): its correlation (double) and d corr / d Y of its entries
__global__ __launch_bounds__(256) void k_stoi_corr_win(const float *Ot, const float *Op, const int *nk_all, int Tm, int Mmax, double *cw, float *dw) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= kBands * Mmax) return;
    const int band = p / Mmax, m = p - band * Mmax;
    const int T = nk_all[b] + 2;
    int Mi, n;
    stoi_windows(T, Tm, Mi, n);
    double corr = 0.0;
    float *dy = dw + (((long)b * kBands + band) * Mmax + m) * kSeg;
    if (m < Mi && n > 0) {
        const float *ot = Ot + (long)b * Tm * kBands, *op = Op + (long)b * Tm * kBands;
        float X[kSeg], Y[kSeg];
#pragma unroll
        for (int i = 0; i < kSeg; i++) {
            const long r = (long)min(m + i, Tm - 1) * kBands + band;
            X[i] = i < n ? ot[r] : 0.0f;
            Y[i] = i < n ? op[r] : 0.0f;
        }
        corr = stoi_window(X, Y, n, (double)max(n, 1), dy);
    } else {
        for (int i = 0; i < kSeg; i++) dy[i] = 0.0f;
    }
    cw[((long)b * kBands + band) * Mmax + m] = corr;
}

// one workgroup per utterance: D[b] = fixed-order sum of its windows' correlations / (15 Mi)
__global__ __launch_bounds__(1024) void k_stoi_corr_sum(const double *cw, const int *nk_all, int Tm, int Mmax, float *D) {
    __shared__ double red[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nk = nk_all[b], T = nk + 2, Ls = kHop * (nk + 1);
    int Mi, n;
    stoi_windows(T, Tm, Mi, n);
    double acc = 0;
    for (int p = tid; p < kBands * Mmax; p += 1024) acc += cw[(long)b * kBands * Mmax + p];  // windows beyond Mi hold 0
    acc = block_sum_d(acc, red, 16) / (15.0 * (double)max(Mi, 1));
    if (tid == 0) D[b] = Ls <= 512 ? 0.99f : (float)acc;
}

// grid (ceil(Tm 15 / 256), B): dOp[b][t][band] = gD[b] / (15 Mi) * sum over the windows m that contain frame t of their d corr / d Y[t - m]
__global__ __launch_bounds__(256) void k_stoi_corr_bwd(const float *dw, const int *nk_all, int Tm, int Mmax, const float *gD, float *dOp) {
    const int b = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Tm * kBands) return;
    const int t = q / kBands, band = q - t * kBands;
    const int nk = nk_all[b], T = nk + 2, Ls = kHop * (nk + 1);
    int Mi, n;
    stoi_windows(T, Tm, Mi, n);
    float d = 0.0f;
    if (Ls > 512 && t < T) {
        const float *w = dw + ((long)b * kBands + band) * Mmax * kSeg;
        const int m_lo = max(0, t - (n - 1)), m_hi = min(t, Mi - 1);
        for (int m = m_lo; m <= m_hi; m++) d += w[(long)m * kSeg + (t - m)];
        d *= gD[b] / (15.0f * (float)max(Mi, 1));
    }
    dOp[(long)b * Tm * kBands + q] = d;
}

// grid (Tm, B): dfr[b][t][j] = Hann[j] * d / d x_j of the frame's envelope terms
__global__ __launch_bounds__(256) void k_stoi_env_bwd(const float *Op, const float *Sp, const float *dOp, const int *nk_all, int Tm, const float *hann_per,
                                                      const int *band_lo, const int *band_hi, float *dfr) {
    __shared__ float2 tw[512];
    __shared__ float2 ds[kBins + 3];  // dpw_k * 2 * (Re_k, Im_k)
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int T = nk_all[b] + 2;
    float *out = dfr + ((long)b * Tm + t) * kWin;
    if (t >= T) { out[tid] = 0.0f; return; }
    for (int i = tid; i < 512; i += 256) { float sn, cs; sincosf(2.0f * kPi * (float)i / 512.0f, &sn, &cs); tw[i] = make_float2(cs, sn); }
    const float *o = Op + ((long)b * Tm + t) * kBands, *dO = dOp + ((long)b * Tm + t) * kBands, *sp = Sp + ((long)b * Tm + t) * kBins * 2;
    for (int k = tid; k < kBins; k += 256) {
        float dpw = 0.0f;
        for (int bd = 0; bd < kBands; bd++)
            if (k >= band_lo[bd] && k < band_hi[bd]) dpw += dO[bd] * 0.5f / o[bd];  // O = sqrt(sum + 1e-14) > 0
        ds[k] = make_float2(2.0f * dpw * sp[2 * k], 2.0f * dpw * sp[2 * k + 1]);
    }
    __syncthreads();
    float acc = 0.0f;
    const int j = tid;
    int idx = 0;  // (j + 128) k mod 512 at k = 0
    const int step = (j + kHop) & 511;
    for (int k = 0; k < kBins; k++) {
        const float2 c = tw[idx], d = ds[k];
        acc += d.x * c.x - d.y * c.y;  // Re_k = sum x cos, Im_k = - sum x sin
        idx = (idx + step) & 511;
    }
    out[j] = hann_per[j] * acc;
}

// dp10[b][n] = sum over the kept frames sample n belongs to of Hann_sym * d st[position], d st gathered through the reflect padding
__global__ __launch_bounds__(256) void k_stoi_gather_bwd(const float *dfr, const int *rank_all, const int *nk_all, const int *n10, long Lo, int Fu, int Tm,
                                                         const float *hann_sym, float *dp10) {
    const int b = blockIdx.y;
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    if (n >= Lo) return;
    const int nk = nk_all[b], T = nk + 2, Ls = kHop * (nk + 1);
    const int *rank = rank_all + (long)b * Fu;
    const float *df = dfr + (long)b * Tm * kWin;
    float acc = 0.0f;
    if (n < n10[b]) {
        const int f1 = (int)(n >> 7);
        for (int f = f1; f >= f1 - 1 && f >= 0; f--) {
            if (f >= Fu) continue;
            const int k = rank[f];
            if (k < 0) continue;
            const int r = (int)(n - ((long)f << 7));
            if (r >= kWin) continue;
            const int s = (k << 7) + r;  // position in the silence-removed signal
            float dst = 0.0f;
            int cand[3] = {s, (s >= 1 && s <= kHop) ? -s : INT32_MIN, (s >= Ls - kHop - 1 && s <= Ls - 2) ? 2 * (Ls - 1) - s : INT32_MIN};
            for (int c = 0; c < 3; c++) {
                const int np = cand[c];
                if (np == INT32_MIN) continue;
                if (c == 2 && (np < Ls || np > Ls + kHop - 1)) continue;
                const int t1 = (np + kHop) >> 7;
                for (int t = t1; t >= t1 - 1 && t >= 0; t--) {
                    const int j = np - (t * kHop - kHop);
                    if (t < T && j >= 0 && j < kWin) dst += df[(long)t * kWin + j];
                }
            }
            acc += hann_sym[r] * dst;
        }
    }
    dp10[(long)b * Lo + n] = acc;
}

__global__ __launch_bounds__(256) void k_stoi_resample_bwd(const float *dp10, const int64_t *lens, long L, long Lo, const float *w, const int *first, int W,
                                                           float *dx) {
    const int b = blockIdx.y;
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m >= L) return;
    const long len = lens[b] < L ? (lens[b] > 0 ? lens[b] : 0) : L;
    float v = 0.0f;
    if (m < len) {
        const int nout = resample_count(len);
        const float *g = dp10 + (long)b * Lo;
        double acc = 0.0;
        for (int i = 0; i < kPhases; i++) {
            const float *wi = w + (long)i * W;
            for (int j = 0; j < W; j++) {
                const long d = m - first[i] - j;
                if (d < 0 || (d & (kStride - 1))) continue;
                const long n = (d >> 3) * kPhases + i;
                if (n < nout) acc += (double)wi[j] * (double)g[n];
            }
        }
        v = (float)acc;
    }
    dx[(long)b * L + m] = v;
}

thread_local char g_stoi_err[256] = "";
int stoi_fail(const char *msg) { snprintf(g_stoi_err, sizeof g_stoi_err, "%s", msg); return SE_ERR_ARG; }

}  // namespace

extern "C" {

const char *se_loss_stoi_last_error(void) { return g_stoi_err; }

int64_t se_loss_stoi_ws_floats(int batch, int64_t length) { return batch > 0 && length > 0 ? (int64_t)stoi_shape(batch, (long)length).total : 0; }

// tables (device): rs_w [5][W] float, rs_first [5] int, hann_sym [256] (np.hanning(256)), hann_per [256] (periodic Hann), band_lo / band_hi [15] int
int se_loss_stoi_fwd(const float *clean, const float *pred, const int64_t *lens, int batch, int64_t length, const float *rs_w, const int *rs_first, int W,
                     const float *hann_sym, const float *hann_per, const int *band_lo, const int *band_hi, float *ws, float *D, void *stream) {
    if (!clean || !pred || !lens || !rs_w || !rs_first || !hann_sym || !hann_per || !band_lo || !band_hi || !ws || !D || batch <= 0 || length <= 0 || W <= 0)
        return stoi_fail("se_loss_stoi_fwd: bad argument");
    const StoiShape s = stoi_shape(batch, (long)length);
    if (s.Fu <= 0) return stoi_fail("se_loss_stoi_fwd: utterances shorter than one 256-sample frame at 10 kHz");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int *n10 = reinterpret_cast<int *>(ws + s.n10), *nk = reinterpret_cast<int *>(ws + s.nk);
    int *order = reinterpret_cast<int *>(ws + s.order), *rank = reinterpret_cast<int *>(ws + s.rank);
    const dim3 gr((unsigned)((s.Lo + 255) / 256), batch);
    hipLaunchKernelGGL(k_stoi_resample, gr, dim3(256), 0, st, clean, lens, (long)length, s.Lo, rs_w, rs_first, W, ws + s.t10, n10);
    hipLaunchKernelGGL(k_stoi_resample, gr, dim3(256), 0, st, pred, lens, (long)length, s.Lo, rs_w, rs_first, W, ws + s.p10, (int *)nullptr);
    hipLaunchKernelGGL(k_stoi_frames, dim3(batch), dim3(1024), (size_t)s.Fu * sizeof(float), st, ws + s.t10, n10, s.Lo, s.Fu, hann_sym, order, rank, nk);
    hipLaunchKernelGGL(k_stoi_env, dim3(s.Tm, batch), dim3(256), 0, st, ws + s.t10, s.Lo, order, nk, s.Fu, s.Tm, hann_sym, hann_per, band_lo, band_hi, ws + s.Ot,
                       (float *)nullptr);
    hipLaunchKernelGGL(k_stoi_env, dim3(s.Tm, batch), dim3(256), 0, st, ws + s.p10, s.Lo, order, nk, s.Fu, s.Tm, hann_sym, hann_per, band_lo, band_hi, ws + s.Op,
                       ws + s.Sp);
    hipLaunchKernelGGL(k_stoi_corr_win, dim3((kBands * s.Mmax + 255) / 256, batch), dim3(256), 0, st, ws + s.Ot, ws + s.Op, nk, s.Tm, s.Mmax,
                       reinterpret_cast<double *>(ws + s.cw), ws + s.dw);
    hipLaunchKernelGGL(k_stoi_corr_sum, dim3(batch), dim3(1024), 0, st, reinterpret_cast<const double *>(ws + s.cw), nk, s.Tm, s.Mmax, D);
    return hipGetLastError() == hipSuccess ? SE_OK : stoi_fail("se_loss_stoi_fwd: launch failed");
}

// ws = the workspace se_loss_stoi_fwd filled for the same inputs; gD [batch] = d loss / d D; dpred [batch][length]
int se_loss_stoi_bwd(const float *gD, const int64_t *lens, int batch, int64_t length, const float *rs_w, const int *rs_first, int W, const float *hann_sym,
                     const float *hann_per, const int *band_lo, const int *band_hi, float *ws, float *dpred, void *stream) {
    if (!gD || !lens || !rs_w || !rs_first || !hann_sym || !hann_per || !band_lo || !band_hi || !ws || !dpred || batch <= 0 || length <= 0 || W <= 0)
        return stoi_fail("se_loss_stoi_bwd: bad argument");
    const StoiShape s = stoi_shape(batch, (long)length);
    if (s.Fu <= 0) return stoi_fail("se_loss_stoi_bwd: utterances shorter than one 256-sample frame at 10 kHz");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int *n10 = reinterpret_cast<int *>(ws + s.n10), *nk = reinterpret_cast<int *>(ws + s.nk), *rank = reinterpret_cast<int *>(ws + s.rank);
    hipLaunchKernelGGL(k_stoi_corr_bwd, dim3((s.Tm * kBands + 255) / 256, batch), dim3(256), 0, st, ws + s.dw, nk, s.Tm, s.Mmax, gD, ws + s.dOp);
    hipLaunchKernelGGL(k_stoi_env_bwd, dim3(s.Tm, batch), dim3(256), 0, st, ws + s.Op, ws + s.Sp, ws + s.dOp, nk, s.Tm, hann_per, band_lo, band_hi, ws + s.dfr);
    hipLaunchKernelGGL(k_stoi_gather_bwd, dim3((unsigned)((s.Lo + 255) / 256), batch), dim3(256), 0, st, ws + s.dfr, rank, nk, n10, s.Lo, s.Fu, s.Tm, hann_sym,
                       ws + s.dp10);
    hipLaunchKernelGGL(k_stoi_resample_bwd, dim3((unsigned)((length + 255) / 256), batch), dim3(256), 0, st, ws + s.dp10, lens, (long)length, s.Lo, rs_w, rs_first, W,
                       dpred);
    return hipGetLastError() == hipSuccess ? SE_OK : stoi_fail("se_loss_stoi_bwd: launch failed");
}

}  // extern "C"
