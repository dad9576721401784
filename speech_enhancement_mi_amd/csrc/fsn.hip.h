// fsn.hip.h - kernels of the FullSubNet path (reference fullsubnet.py:685-961).
//
//   k_fsn_mag        |X| of every microphone into the full-band LSTM's input layout [B][T][Kp] (+ partial sums)
//   k_fsn_scale      CumLayerNorm: x /= running_mean + eps, elementwise (fullsubnet.py:184-201)
//   k_fsn_runmean    per-stream running-mean update from the partial sums (one thread per stream)
//   k_lstm_step_x6   one LSTM time step for R rows as a fused GEMM: [x_t | h_{t-1}] . [W_ih | W_hh]^T + b -> gates -> (c, h),
//                    fp32-accurate bf16x6 MFMA (see gemm.hip.h); the four gate columns of a hidden unit sit in the same
//                    lane, so the cell update is the epilogue.  Used for both the full-band (R = B) and the sub-band
//                    (R = B*F = 51 456 at B = 256) models; the latter is 99 % of FullSubNet's 15.5 GFLOP per frame.
//   k_fsn_unfold     sub-band input: 31 reflect-padded neighbours of the normalised mic-0 magnitude + the full-band
//                    output, laid out time-major [T][B*F][32] so each step's GEMM operand is contiguous (+ partial sums)
//   k_fsn_sbfc       Linear(384 -> 2) of the sub-band model for one time step
//   k_fsn_mask       decompress_cIRM + complex multiply with the mic-0 spectrum (fullsubnet.py:949-953)
#pragma once
#include <hip/hip_runtime.h>
#include "gemm.hip.h"
#include "norm.hip.h"

namespace se {

// ---- magnitude + partial sums -----------------------------------------------------------------------------------------
struct FsnMagArgs {
    const float *re, *im; // element (b, m, t, f) at ptr[b*sB + m*sM + t*sT + f*sF] (float units): interleaved complex spectra
                          // (im = re + 1) and the reference's planar [B, 2M, F, T] input (fullsubnet.py:835-844) both fit
    long sB, sM, sT, sF;
    float *mag;           // [B][T][Kp], column m*F + f; columns >= M*F stay zero
    float *partial;       // [B][gridDim.x] block sums
    int M, T, F, Kp;
};

__global__ __launch_bounds__(256) void k_fsn_mag(FsnMagArgs a) {
    __shared__ double red[4];
    const int b = blockIdx.y;
    const int n = a.M * a.T * a.F;
    float part = 0.0f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int m = i / (a.T * a.F), r = i - m * a.T * a.F, t = r / a.F, f = r - t * a.F;
        const long off = (long)b * a.sB + (long)m * a.sM + (long)t * a.sT + (long)f * a.sF;
        const float vx = a.re[off], vy = a.im[off];
        const float mg = sqrtf(vx * vx + vy * vy + kEps);  // fullsubnet.py:782
        a.mag[((long)b * a.T + t) * a.Kp + m * a.F + f] = mg;
        part += mg;
    }
    const double s = block_sum((double)part, red);
    if (threadIdx.x == 0) a.partial[(long)b * gridDim.x + blockIdx.x] = (float)s;
}

// run_mean[b] <- first call ? mean : alpha*run_mean + (1-alpha)*mean ; scale[b] = run_mean + eps
__global__ void k_fsn_runmean(const float *partial, int nslot, double count, float *run_mean, float *denom, int B, int first, float alpha) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s = 0;
    for (int i = 0; i < nslot; i++) s += (double)partial[(long)b * nslot + i];
    const float mean = (float)(s / count);
    const float rm = first ? mean : alpha * run_mean[b] + (1.0f - alpha) * mean;
    run_mean[b] = rm;
    denom[b] = rm + kEps;
}

__global__ void k_fsn_scale(float *x, long per_stream, const float *denom) {
    const int b = blockIdx.y;
    const float d = denom[b];
    float *p = x + (long)b * per_stream;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_stream; i += (long)gridDim.x * blockDim.x) p[i] = p[i] / d;
}

// ---- sub-band unfold: sbin[t][(b, f)][W + 1] ------------------------------------------------------------------------------
struct FsnUnfoldArgs {
    const float *mag;     // normalised [B][T][Kp], mic 0 in columns [0, F)
    const float *fb_out;  // [B*T][F]  (row b*T + t)
    float *sbin;          // [T][B*F][SI]
    float *partial;       // [B][gridDim.x]
    int B, T, F, Kp, NB, SI;
};

__global__ __launch_bounds__(256) void k_fsn_unfold(FsnUnfoldArgs a) {
    __shared__ double red[4];
    const int b = blockIdx.y;
    const int n = a.T * a.F * a.SI;
    const int W = 2 * a.NB + 1;
    float part = 0.0f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int t = i / (a.F * a.SI), r = i - t * a.F * a.SI, f = r / a.SI, j = r - f * a.SI;
        float v;
        if (j < W) {
            int fi = f - a.NB + j;  // reflect pad (functional.pad mode="reflect", fullsubnet.py:320)
            if (fi < 0) fi = -fi;
            if (fi >= a.F) fi = 2 * (a.F - 1) - fi;
            v = a.mag[((long)b * a.T + t) * a.Kp + fi];
        } else {
            v = a.fb_out[((long)b * a.T + t) * a.F + f];
        }
        a.sbin[((long)t * a.B * a.F + (long)b * a.F + f) * a.SI + j] = v;
        part += v;
    }
    const double s = block_sum((double)part, red);
    if (threadIdx.x == 0) a.partial[(long)b * gridDim.x + blockIdx.x] = (float)s;
}

// sbin is time-major, so the per-stream scale cannot use k_fsn_scale's contiguous layout
__global__ void k_fsn_scale_sb(float *sbin, int B, int T, int F, int SI, const float *denom) {
    const long per_t = (long)B * F * SI, total = per_t * T;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i % per_t;
        const int b = (int)(r / ((long)F * SI));
        sbin[i] = sbin[i] / denom[b];
    }
}

// ---- fused LSTM step -----------------------------------------------------------------------------------------------------
struct LstmStepArgs {
    const float *x;    // [R][ldx] input of this step (K1 valid columns)
    long ldx;
    int K1, K1p;       // K1p = K1 rounded up to a multiple of 32 (weight planes are padded with zeros)
    const float *hprev;  // [R][H]
    const __bf16 *Wp;  // [3][4H][K1p + Hp] planes of [W_ih | W_hh], Hp = H rounded up to 32
    const float *bias; // [4H] = b_ih + b_hh
    float *c;          // [R][H] cell state, updated in place
    float *hout;       // [R][H]
    float *hseq;       // optional second copy of h (row stride ldseq), nullptr = off
    long ldseq;
    int R, H;
#ifdef SE_LSTM_STAMPS
    unsigned long long *stamps;  // diagnostic build only: per-segment cycle sums of one wave (never read by the kernel)
#endif
};

#ifdef SE_LSTM_STAMPS
#define SE_STAMP(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#endif

// PL = operand planes: 3 = fp32-accurate (six products), 2 = "bf16x3" (hi, mid: three products, fsn_config.precision = 2)
// (A variant that staged [x_t | h_{t-1}] from pre-split planes written by the producers - a pure copy, no VALU split in the loop - was
//  7 % / 4 % slower: 1.5x the operand bytes, and the split was never the wait, see k_lstm_step_big's header.)
template <int PL>
__global__ __launch_bounds__(256) void k_lstm_step_x6(LstmStepArgs a) {
    __shared__ __align__(16) __bf16 Ap[PL][kGemmBM * kXLd];
    __shared__ __align__(16) __bf16 Wl[PL][kGemmBN * kXLd];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.y * kGemmBM, h0 = blockIdx.x * 32;
    const int H = a.H, Hp = (H + 31) & ~31, Kt = a.K1p + Hp;
    const int nck = Kt / kGemmKC;
    f32x16 acc[4];
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[g][r] = 0.0f;

    f32x4 qa[4];
    uint4 qw[2 * PL];
    auto issue = [&](int ck) {
        const int k0 = ck * kGemmKC;
        const bool from_x = k0 < a.K1p;
        const float *src = from_x ? a.x : a.hprev;
        const long ld = from_x ? a.ldx : (long)H;
        const int kbase = from_x ? k0 : k0 - a.K1p, kval = from_x ? a.K1 : H;
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int slot = tid + it * 256, r = slot >> 3, kq = (slot & 7) * 4;
            const int row = min(m0 + r, a.R - 1), k = min(kbase + kq, kval - 4);  // K1, H are multiples of 4 (host-checked)
            qa[it] = *reinterpret_cast<const f32x4 *>(src + (long)row * ld + k);
        }
#pragma unroll
        for (int it = 0; it < 2 * PL; it++) {
            const int seg = tid + it * 256, plane = seg >> 9, w = seg & 511, r = w >> 2, q = (w & 3) * 8;
            const int hid = min(h0 + (r & 31), H - 1), wrow = (r >> 5) * H + hid;
            qw[it] = *reinterpret_cast<const uint4 *>(a.Wp + ((long)plane * 4 * H + wrow) * Kt + k0 + q);
        }
    };
    issue(0);
#ifdef SE_LSTM_STAMPS
    unsigned long long ts0, ts1, ts2, ts3, ts4, ts5, sum[5] = {0, 0, 0, 0, 0};
#endif
    for (int ck = 0; ck < nck; ck++) {
        const int k0 = ck * kGemmKC;
        const bool from_x = k0 < a.K1p;
        const int kbase = from_x ? k0 : k0 - a.K1p, kval = from_x ? a.K1 : H;
#ifdef SE_LSTM_STAMPS
        SE_STAMP(ts0);
#endif
        __syncthreads();
#ifdef SE_LSTM_STAMPS
        SE_STAMP(ts1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SE_STAMP(ts2);
#endif
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int slot = tid + it * 256, r = slot >> 3, kq = (slot & 7) * 4;
            const bool ok = (m0 + r < a.R) && (kbase + kq < kval);  // 4-aligned windows are either all valid or all padding
            bf16x4 h, m, l;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                __bf16 hh, mm, ll;
                split3(ok ? qa[it][e] : 0.0f, hh, mm, ll);
                h[e] = hh; m[e] = mm; l[e] = ll;
            }
            *reinterpret_cast<bf16x4 *>(&Ap[0][r * kXLd + kq]) = h;
            *reinterpret_cast<bf16x4 *>(&Ap[1][r * kXLd + kq]) = m;
            if (PL > 2) *reinterpret_cast<bf16x4 *>(&Ap[PL - 1][r * kXLd + kq]) = l;
        }
#pragma unroll
        for (int it = 0; it < 2 * PL; it++) {
            const int seg = tid + it * 256, plane = seg >> 9, w = seg & 511, r = w >> 2, q = (w & 3) * 8;
            const bool ok = h0 + (r & 31) < H;
            *reinterpret_cast<uint4 *>(&Wl[plane][r * kXLd + q]) = ok ? qw[it] : make_uint4(0, 0, 0, 0);
        }
#ifdef SE_LSTM_STAMPS
        SE_STAMP(ts3);
#endif
        __syncthreads();
#ifdef SE_LSTM_STAMPS
        SE_STAMP(ts4);
#endif
        if (ck + 1 < nck) issue(ck + 1);
#pragma unroll
        for (int ks = 0; ks < kGemmKC; ks += 16) {
            bf16x8 fa[PL];
#pragma unroll
            for (int p = 0; p < PL; p++) fa[p] = *reinterpret_cast<const bf16x8 *>(&Ap[p][(wave * 32 + l31) * kXLd + ks + half * 8]);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                bf16x8 fb[PL];
#pragma unroll
                for (int p = 0; p < PL; p++) fb[p] = *reinterpret_cast<const bf16x8 *>(&Wl[p][(g * 32 + l31) * kXLd + ks + half * 8]);
                f32x16 c = acc[g];
                if (PL > 2) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[PL - 1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PL - 1], fb[0], c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], c, 0, 0, 0);
                acc[g] = c;
            }
        }
#ifdef SE_LSTM_STAMPS
        SE_STAMP(ts5);  // MFMAs issued (not retired): the tail shows up in the next iteration's barrier segment
        sum[0] += ts1 - ts0; sum[1] += ts2 - ts1; sum[2] += ts3 - ts2; sum[3] += ts4 - ts3; sum[4] += ts5 - ts4;
#endif
    }
#ifdef SE_LSTM_STAMPS
    if (a.stamps && blockIdx.x == 5 && blockIdx.y == gridDim.y / 2 && tid == 64) {
        for (int i = 0; i < 5; i++) a.stamps[i] = sum[i];
        a.stamps[5] = nck;
    }
#endif
    // epilogue: torch.nn.LSTM cell, gate order i, f, g, o
    const int j = h0 + l31;
    if (j >= H) return;
    const float bi = a.bias[j], bf = a.bias[H + j], bg = a.bias[2 * H + j], bo = a.bias[3 * H + j];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m >= a.R) continue;
        const float ig = 1.0f / (1.0f + expf(-(acc[0][r] + bi)));
        const float fg = 1.0f / (1.0f + expf(-(acc[1][r] + bf)));
        const float gg = tanhf(acc[2][r] + bg);
        const float og = 1.0f / (1.0f + expf(-(acc[3][r] + bo)));
        const long idx = (long)m * H + j;
        const float cn = fg * a.c[idx] + ig * gg;
        const float hn = og * tanhf(cn);
        a.c[idx] = cn;
        a.hout[idx] = hn;
        if (a.hseq) a.hseq[(long)m * a.ldseq + j] = hn;
    }
}

// ---- the same step on a 256-row x 64-unit (256 gate columns) tile, 8 waves ---------------------------------------------------------
// In-kernel stamps of k_lstm_step_x6 at the sub-band model's shape (R = B*F = 51 456, H = 384; fp32 mode, per 32-deep chunk and wave):
// 1 600 cycles staging (split + ds_write), 3 000 cycles for 48 MFMAs (1 536 of matrix-pipe time: the B fragments are read just in
// time, and the other workgroup's staging shares the SIMD's issue slots), 500 in barriers; the global loads are hidden (100 cycles).
// (Issuing the loads two chunks ahead from a second register set changes nothing: -2 %.)  Per MFMA the 128 x 128 tile pays too much around it.  Here a workgroup owns 256 rows x 64 units: wave w = rows (w & 3) * 64 .. + 64
// (two 32-row MFMA tiles) x units (w >> 2) * 32 .. + 32, the four gates of a unit in the same lane (acc[rt][gate]): 96 MFMAs per
// chunk against the same 16 split values per lane and 18 instead of 30 fragment reads per 48, half the operand re-reads out of L2.
//   * W never touches registers: each wave issues 2 * PL `buffer_load ... lds` of 1 KB (16 weight rows x 64 B) per chunk into a
//     double-buffered stage, one chunk ahead (a register-staged W made hipcc sink the loads to the end of the iteration, exposed).
//   * LDS rows are 64 B (32 bf16) unpadded; 16-B slot s of row r is stored at slot s ^ ((r >> 2) & 3): a ds_read_b128 lane group
//     (16 rows distinct mod 16, one k slot) covers all 16 slots of the 256-B bank row, conflict-free.  LDS-DMA destinations are
//     lane-linear, so the swizzle is applied to the lane's SOURCE k offset.
//   * The fp32 -> split-bf16 conversion of chunk ck + 1 happens in registers between the two MFMA groups of chunk ck; the store phase
//     between the barriers is 4 * PL ds_write_b64 per lane.
// LDS: (2 W stages + 1 A stage) * PL * 16 KB = 144 KB (fp32 mode): one workgroup of 8 waves per CU.
constexpr int kLbM = 256, kLbU = 64, kLbPlane = kLbM * kGemmKC;  // bf16 elements per operand plane (16 KB)
template <int PL>
__global__ __launch_bounds__(512) void k_lstm_step_big(LstmStepArgs a) {
    extern __shared__ __align__(16) unsigned char lstm_lds[];
    __bf16 *Wst = reinterpret_cast<__bf16 *>(lstm_lds);            // [2][PL][256 rows][32]
    __bf16 *Ast = Wst + 2 * PL * kLbPlane;                          // [1 or 2][PL][256 rows][32]
    // bf16x3 (PL = 2): the A stage is double-buffered too (128 KB in all), the stores of chunk ck + 1 go out between the MFMA groups of
    // chunk ck and ONE barrier per chunk is left; with three planes a second A stage does not fit (192 KB)
    constexpr bool kOneBarrier = PL == 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = (wave & 3) * 64, wn = (wave >> 2) * 32;
    const int m0 = blockIdx.y * kLbM, h0 = blockIdx.x * kLbU;
    const int H = a.H, Hp = (H + 31) & ~31, Kt = a.K1p + Hp;
    const int nck = Kt / kGemmKC;
    f32x16 acc[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[rt][g][r] = 0.0f;

    // W by LDS-DMA: instruction j of this wave fills plane j >> 1, rows (wave + 8 (j & 1)) * 16 .. + 16; lane i -> row + (i >> 2), physical slot i & 3
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(a.Wp), 0, (unsigned)((long)PL * 4 * H * Kt * 2), 0x00020000);
    unsigned woff[2 * PL];
#pragma unroll
    for (int j = 0; j < 2 * PL; j++) {
        const int r = (wave + 8 * (j & 1)) * 16 + (lane >> 2);
        const int ks = (lane & 3) ^ ((r >> 2) & 3);
        const int wrow = (r >> 6) * H + h0 + (r & 63);
        woff[j] = (unsigned)((((long)(j >> 1) * 4 * H + wrow) * Kt + ks * 8) * 2);
    }
    auto dma_w = [&](int ck) {
        __bf16 *dst = Wst + (ck & 1) * PL * kLbPlane;
#pragma unroll
        for (int j = 0; j < 2 * PL; j++) {
            const unsigned v = woff[j];  // a local: with the array element as the argument hipcc's host pass drops the kernel's stub without a diagnostic
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void *)(dst + (j >> 1) * kLbPlane + (wave + 8 * (j & 1)) * 16 * kGemmKC), 16,
                                                     v, ck * kGemmKC * 2, 0, 0);
        }
    };
    f32x4 qa[4];
    bf16x4 pa[4][PL];
    auto issue = [&](int ck) {
        const int k0 = ck * kGemmKC;
        const bool from_x = k0 < a.K1p;
        const float *src = from_x ? a.x : a.hprev;
        const long ld = from_x ? a.ldx : (long)H;
        const int kbase = from_x ? k0 : k0 - a.K1p, kval = from_x ? a.K1 : H;
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int slot = tid + it * 512, r = slot >> 3, kq = (slot & 7) * 4;
            const int row = min(m0 + r, a.R - 1), k = min(kbase + kq, kval - 4);  // K1, H are multiples of 4 (host-checked)
            qa[it] = *reinterpret_cast<const f32x4 *>(src + (long)row * ld + k);
        }
    };
    auto split = [&](int ck) {
        const int k0 = ck * kGemmKC;
        const bool from_x = k0 < a.K1p;
        const int kbase = from_x ? k0 : k0 - a.K1p, kval = from_x ? a.K1 : H;
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int slot = tid + it * 512, r = slot >> 3, kq = (slot & 7) * 4;
            const bool ok = (m0 + r < a.R) && (kbase + kq < kval);  // 4-aligned windows are either all valid or all padding
#pragma unroll
            for (int e = 0; e < 4; e++) {
                __bf16 hh, mm, ll;
                split3(ok ? qa[it][e] : 0.0f, hh, mm, ll);
                pa[it][0][e] = hh; pa[it][1][e] = mm;
                if (PL > 2) pa[it][PL - 1][e] = ll;
            }
        }
    };
    const int sw = (l31 >> 2) & 3;  // every fragment row is a multiple of 32 plus l31
    bf16x8 fa[2][PL];
    auto load_fa = [&](int ck, int ks) {
        const __bf16 *Ap = Ast + (kOneBarrier ? (ck & 1) * PL * kLbPlane : 0);
        const int slot = (((ks >> 3) + half) ^ sw) * 8;
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int p = 0; p < PL; p++) fa[rt][p] = *reinterpret_cast<const bf16x8 *>(&Ap[p * kLbPlane + (wm + rt * 32 + l31) * kGemmKC + slot]);
    };
    auto store_a = [&](int ck) {  // the split values of chunk ck (in pa) -> its A stage
        __bf16 *Ap = Ast + (kOneBarrier ? (ck & 1) * PL * kLbPlane : 0);
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int slot = tid + it * 512, r = slot >> 3, kq = (slot & 7) * 4;
            const int off = r * kGemmKC + (((kq >> 3) ^ ((r >> 2) & 3)) * 8) + (kq & 4);
#pragma unroll
            for (int p = 0; p < PL; p++) *reinterpret_cast<bf16x4 *>(&Ap[p * kLbPlane + off]) = pa[it][p];
        }
    };
    auto mfma_gate = [&](int ck, int ks, int g) {  // the 2 x (3 or 6) products of gate g's 32 columns
        const __bf16 *Wl = Wst + (ck & 1) * PL * kLbPlane;
        const int slot = (((ks >> 3) + half) ^ sw) * 8;
        bf16x8 fb[PL];
#pragma unroll
        for (int p = 0; p < PL; p++) fb[p] = *reinterpret_cast<const bf16x8 *>(&Wl[p * kLbPlane + (g * kLbU + wn + l31) * kGemmKC + slot]);
#pragma unroll
        for (int rt = 0; rt < 2; rt++) {
            f32x16 c = acc[rt][g];
            if (PL > 2) {
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rt][1], fb[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rt][0], fb[PL - 1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rt][PL - 1], fb[0], c, 0, 0, 0);
            }
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rt][0], fb[1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rt][1], fb[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rt][0], fb[0], c, 0, 0, 0);
            acc[rt][g] = c;
        }
    };
    dma_w(0);
    issue(0);
    split(0);
    if constexpr (kOneBarrier) {
        store_a(0);
        __builtin_amdgcn_s_waitcnt(0x0070);
        __syncthreads();
        for (int ck = 0; ck < nck; ck++) {
            const int nx = min(ck + 1, nck - 1);  // the last iteration re-fetches its own chunk into the stages nobody reads
            const int nst = ck + 1;               // stage parity of the chunk being prepared (also on the last, unused, round)
            load_fa(ck, 0);
            mfma_gate(ck, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            {   // W(ck + 1) -> W stage (ck + 1) & 1, read last in iteration ck - 1: every wave is past that iteration's barrier
                __bf16 *dst = Wst + (nst & 1) * PL * kLbPlane;
#pragma unroll
                for (int j = 0; j < 2 * PL; j++) {
                    const unsigned v = woff[j];
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void *)(dst + (j >> 1) * kLbPlane + (wave + 8 * (j & 1)) * 16 * kGemmKC), 16,
                                                             v, nx * kGemmKC * 2, 0, 0);
                }
            }
            issue(nx);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 1; g < 4; g++) mfma_gate(ck, 0, g);
            split(nx);
            store_a(nst);
            load_fa(ck, 16);
#pragma unroll
            for (int g = 0; g < 4; g++) mfma_gate(ck, 16, g);
            __builtin_amdgcn_s_waitcnt(0x0070);  // this wave's W(ck + 1) DMA and A(ck + 1) stores have landed
            __syncthreads();
        }
    } else {
#ifdef SE_LSTM_STAMPS
    unsigned long long ts0, ts1, ts2, ts3, ts4, ts5, ts6, sum[6] = {0, 0, 0, 0, 0, 0};
#define SE_BSTAMP(t) SE_STAMP(t)
#else
#define SE_BSTAMP(t)
#endif
    for (int ck = 0; ck < nck; ck++) {
        SE_BSTAMP(ts0);
        __syncthreads();  // every wave is done with A(ck - 1) and W stage (ck + 1) & 1
        SE_BSTAMP(ts1);
        store_a(ck);
        __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): this wave's W(ck) DMA (issued one iteration ago) and A stores have landed
        SE_BSTAMP(ts2);
        __syncthreads();
        SE_BSTAMP(ts3);
        const int nx = min(ck + 1, nck - 1);  // the last iteration re-fetches its own chunk (unused): the loop body stays one basic block
        // the matrix pipe first: all eight waves leave the barrier together, and 10 x 1 KB of loads per wave queue for ~900 cycles at the
        // CU's one texture-address unit; behind the first gate's MFMAs that wait is in the pipe's shadow
        load_fa(ck, 0);
        mfma_gate(ck, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        SE_BSTAMP(ts4);
        dma_w(nx + (nx == ck ? 1 : 0));       // ... into the stage nobody reads
        issue(nx);
        __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise sinks the operand loads to the end of the iteration (register pressure) and waits on them at once
        SE_BSTAMP(ts5);
#pragma unroll
        for (int g = 1; g < 4; g++) mfma_gate(ck, 0, g);
        split(nx);
        load_fa(ck, 16);
#pragma unroll
        for (int g = 0; g < 4; g++) mfma_gate(ck, 16, g);
        SE_BSTAMP(ts6);
#ifdef SE_LSTM_STAMPS
        sum[0] += ts1 - ts0; sum[1] += ts2 - ts1; sum[2] += ts3 - ts2; sum[3] += ts4 - ts3; sum[4] += ts5 - ts4; sum[5] += ts6 - ts5;
#endif
    }
#ifdef SE_LSTM_STAMPS
    if (a.stamps && blockIdx.x == 3 && blockIdx.y == gridDim.y / 2 && tid == 64) {
        for (int i = 0; i < 6; i++) a.stamps[i] = sum[i];
        a.stamps[6] = nck;
    }
#endif
    }
    __builtin_amdgcn_s_waitcnt(0x0070);  // drain the dummy DMA before the workgroup's LDS is released
    // epilogue: torch.nn.LSTM cell, gate order i, f, g, o
    const int j = h0 + wn + l31;
    const float bi = a.bias[j], bf = a.bias[H + j], bg = a.bias[2 * H + j], bo = a.bias[3 * H + j];
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int m = m0 + wm + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m >= a.R) continue;
            const float ig = 1.0f / (1.0f + expf(-(acc[rt][0][r] + bi)));
            const float fg = 1.0f / (1.0f + expf(-(acc[rt][1][r] + bf)));
            const float gg = tanhf(acc[rt][2][r] + bg);
            const float og = 1.0f / (1.0f + expf(-(acc[rt][3][r] + bo)));
            const long idx = (long)m * H + j;
            const float cn = fg * a.c[idx] + ig * gg;
            const float hn = og * tanhf(cn);
            a.c[idx] = cn;
            a.hout[idx] = hn;
            if (a.hseq) a.hseq[(long)m * a.ldseq + j] = hn;
        }
}

// ---- sub-band Linear(H -> 2) for one time step: one wave per row, lanes across k ---------------------------------------------
__global__ __launch_bounds__(256) void k_fsn_sbfc(const float *h, const float *w /*[2][H]*/, const float *bias, float *mask /*[R][2][T]*/,
                                                  int R, int H, int T, int t) {
    const int lane = threadIdx.x & 63;
    const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    for (int row = wid; row < R; row += nw) {
        float s0 = 0.0f, s1 = 0.0f;
        for (int k = lane; k < H; k += 64) {
            const float v = h[(long)row * H + k];
            s0 += w[k] * v;
            s1 += w[H + k] * v;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off, 64); s1 += __shfl_down(s1, off, 64); }
        if (lane == 0) {
            mask[((long)row * 2 + 0) * T + t] = s0 + bias[0];
            mask[((long)row * 2 + 1) * T + t] = s1 + bias[1];
        }
    }
}

}  // namespace se

#include "fsn_mask.hip.h"
