// train_ops.inc.h - hand-written building blocks of the TRAINING step (reference train.py:195-204 runs
// torch autograd over CRN.py:290-401 / 196-287); included at the end of se_engine.hip, exported through the se_train_* C ABI.
//
// What runs where in `TrainableCRN` with use_hip_kernels(True) (speech_enhancement_mi_amd/train_ops.py):
//   convolution forward            k_conv_igemm (fp32-exact MFMA implicit GEMM, conv_igemm.hip.h) on [B][C][T][F] activations
//   conv / deconv input gradient   the SAME kernel with the roles swapped: d/dx of the strided causal convolution is the
//                                  "keep the last T" transposed convolution of dy (and vice versa) - identical index algebra
//   conv / deconv weight gradient  k_corr_wgrad below: C[a][b][kf][kt] = sum_{batch,t,m} G[a][t][m] * S[b][t-(2-kt)d][2m+kf-2]
//   dense layers (gi, fc, dW, dx)  k_gemm_skinny below (fp32-exact MFMA GEMM, 32 x 32 tiles, K split over the four waves)
//   GRU forward step               k_gru_step with the gate values saved for the backward pass
//   GRU backward step (BPTT)       k_gru_bwd_gates (pointwise gate derivatives) + k_gemm_skinny (dh_{t-1} += dgh W_hh)
// Everything is fp32 with exact MFMA accumulation order (v_mfma_f32_32x32x2_f32 / 16x16x4), so gradients agree with torch
// autograd to rounding; the checker is tests/test_gpu_round2.py::test_hip_training_ops_vs_autograd.

namespace se {
int train_fail(int code, const char *fmt, ...);  // se_train.hip (owns se_train_last_error)
void launch_arrange_w(const float *w, float *out, long sCo, long sCi, int Co, int Ci, int ntap, int CC, int nchunk, int CoPad, int one_by_one,
                      const int *kf, const int *kt, hipStream_t st);
}
#define tfail se::train_fail

namespace {

struct TrainConvGeo {
    ConvArgs a{};
    int NT = 0, grid_x = 0;
    size_t lds = 0;
    int tap_kf[kMaxTaps]{}, tap_kt[kMaxTaps]{};
};

// kind 0: TemporalConv2d (5x3, stride (2,1), causal dilated, history rows from xprev)   x [B][Ci][T][Fi] -> y [B][Co][T][Fy=Fo]
// kind 1 / 2: even / odd output-frequency parity of TemporalConvTranspose2d (keep the last T columns), y [B][Co][T][Fy]
int train_conv_geometry(int kind, int Ci, int Co, int T, int Fi, int Fy, int dil, TrainConvGeo &g) {
    std::vector<std::array<int, 4>> taps;
    int FP, s, os, oo, colpad, tlo_off, St;
    if (kind == 0) {
        for (int kf = 0; kf < 5; kf++)
            for (int kt = 0; kt < 3; kt++) taps.push_back({kf, kt, kt, kf});
        FP = Fy; s = 2; os = 1; oo = 0; colpad = 2; tlo_off = -2 * dil; St = Fi + 4;
        if (Fy != (Fi - 1) / 2 + 1) return tfail(SE_ERR_ARG, "strided conv: Fy must be (Fi-1)/2+1");
    } else if (kind == 1) {
        for (int kf = 0; kf < 5; kf += 2)
            for (int kt = 0; kt < 3; kt++) taps.push_back({kf, kt, 2 - kt, 2 - kf / 2});
        FP = (Fy + 1) / 2; s = 1; os = 2; oo = 0; colpad = 1; tlo_off = 0; St = Fi + 2;
    } else if (kind == 2) {
        for (int kf = 1; kf < 5; kf += 2)
            for (int kt = 0; kt < 3; kt++) taps.push_back({kf, kt, 2 - kt, 1 + (3 - kf) / 2});
        FP = Fy / 2; s = 1; os = 2; oo = 1; colpad = 1; tlo_off = 0; St = Fi + 2;
    } else if (kind == 3) {  // 1x1 convolution (the decoder's stacked residual / residualmask pair, CRN.py:372-375, and its input gradient)
        taps.push_back({2, 2, 0, 0});
        FP = Fy; s = 1; os = 1; oo = 0; colpad = 0; tlo_off = 0; St = Fi; dil = 0;
        if (Fy != Fi) return tfail(SE_ERR_ARG, "1x1 conv: Fy must equal Fi");
    } else return tfail(SE_ERR_ARG, "unknown conv kind %d", kind);
    if ((kind == 1 || kind == 2) && (Fy < 2 * Fi - 1 || Fy > 2 * Fi)) return tfail(SE_ERR_ARG, "transposed conv: Fy must be 2 Fi - 1 or 2 Fi");
    if (kind != 3 && T <= 2 * dil) return tfail(SE_ERR_ARG, "segment of %d frames is not longer than the dilation history 2 x %d (CRN.py:333-337 keeps older columns then; unsupported)", T, dil);
    const int ntap = (int)taps.size(), ngroup = kind == 3 ? 1 : 3;
    const int P = T * FP, tiles = (P + 31) / 32;
    const int CoPad = (Co + 31) / 32 * 32;
    if (CoPad > 128 || CoPad == 96) return tfail(SE_ERR_ARG, "%d output channels unsupported (32, 64 or 128 GEMM rows)", Co);
    const int MT = CoPad / 32, NCG = 4 / MT;
    const int CiPad = (Ci + 1) / 2 * 2;
    int tpw = 0, NT = 0, n_wg = 0, grouped = 0, Rmax = 0, CC = 0;
    auto bytes = [&](int cc) { return sizeof(float) * ((size_t)ntap * cc * CoPad + (size_t)cc * Rmax * St); };
    auto fits = [&](int cc) {
        return bytes(cc) <= 48 * 1024 && (size_t)cc * Rmax * St <= 256 * kPatchPerThread && (size_t)ntap * cc * CoPad <= 256 * 4 * kWeightPerThread;
    };
    bool ok = false;
    for (int ntmax = 4; ntmax >= 1 && !ok; ntmax--) {  // fewer tiles per workgroup = fewer patch rows, until a 2-channel chunk fits
        n_wg = (tiles + NCG * ntmax - 1) / (NCG * ntmax);
        tpw = (tiles + n_wg - 1) / n_wg;
        NT = (tpw + NCG - 1) / NCG;
        int rows_pos = (tpw * 32 + FP - 1) / FP + 1;
        if (rows_pos > T) rows_pos = T;
        grouped = ngroup * rows_pos < rows_pos + (ngroup - 1) * dil;
        Rmax = grouped ? ngroup * rows_pos : rows_pos + (ngroup - 1) * dil;
        CC = CiPad;
        while (CC > 2 && !fits(CC)) CC -= 2;
        ok = fits(CC);
    }
    if (!ok) return tfail(SE_ERR_ARG, "conv chunk does not fit the staging registers (Fi %d, dilation %d)", Fi, dil);
    int nchunk = (CiPad + CC - 1) / CC;
    CC = ((CiPad + nchunk - 1) / nchunk + 1) / 2 * 2;
    nchunk = (CiPad + CC - 1) / CC;
    ConvArgs &a = g.a;
    a.Ci = Ci; a.Co = Co; a.CoPad = CoPad; a.T = T; a.Fi = Fi; a.FP = FP; a.Fy = Fy;
    a.s = s; a.os = os; a.oo = oo; a.colpad = colpad; a.tlo_off = tlo_off; a.ngroup = ngroup; a.dil = dil; a.grouped = grouped;
    a.ntap = ntap; a.CC = CC; a.nchunk = nchunk; a.tiles_per_wg = tpw; a.St = St;
    a.relu_lo = 0; a.relu_hi = 0; a.act = 0; a.gate_pairs = 0; a.Cy = Co; a.cy0 = 0; a.par_rows = 0; a.gatew = nullptr;
    a.stats = nullptr; a.blend = 0;
    for (int t = 0; t < ntap; t++) { a.rowgrp[t] = taps[t][2]; a.coloff[t] = taps[t][3]; g.tap_kf[t] = taps[t][0]; g.tap_kt[t] = taps[t][1]; }
    g.NT = NT; g.grid_x = n_wg; g.lds = bytes(CC);
    return 0;
}

void train_conv_attributes() {  // large dynamic LDS opt-in of every convolution instance: once per process and device
    static thread_local int done_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (done_dev == dev) return;
    conv_set_attributes();
    done_dev = dev;
}

}  // namespace

namespace se {

// C[a][b][kf][kt] += sum over (batch, t, m) of G[a][t][m] * S[b][t - (2 - kt) d][2 m + kf - 2]   (rows t' < 0 from Sprev, or zero)
//   weight gradient of the strided convolution:   G = dy [B][Co][T][Fo], S = x [B][Ci][T][Fi] (+ history) -> dW [Co][Ci][5][3]
//   weight gradient of the transposed convolution: G = x [B][Ci][T][Fi], S = dy [B][Co][T][2Fi-1]          -> dW [Ci][Co][5][3]
// One workgroup = one 32 x 32 tile of C (rows a, columns n = b * 15 + tap) over a slice of the (batch, t) rows; its four waves
// take rows round-robin, v_mfma_f32_32x32x2_f32 contracts two positions m per step, the waves' tiles are summed through LDS
// and added to C with float atomics (the host zeroes C first).
struct WgradArgs {
    const float *G, *S, *Sprev;
    float *C;
    int B, Ca, Cb, T, Fm, Fs, dil, rows_per_split;
    int ntap, fs;  // 15 taps with frequency stride 2 (5x3 kernels), or ntap = 1, fs = 1 (1x1 convolutions)
    long split_stride;  // > 0: split z writes its partial tile to C + z * split_stride (deterministic two-stage sum); 0: atomics into C
};

__global__ __launch_bounds__(256) void k_corr_wgrad(WgradArgs a) {
    __shared__ float red[3][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int a0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int N = a.Cb * a.ntap;
    const int ai = a0 + l31, n = n0 + l31;
    const bool a_ok = ai < a.Ca, n_ok = n < N;
    const int bch = n_ok ? n / a.ntap : 0, tap = n_ok ? n - bch * a.ntap : 0;
    const int kf = a.ntap == 1 ? 2 : tap / 3, kt = a.ntap == 1 ? 2 : tap - kf * 3;
    const int toff = -(2 - kt) * a.dil;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    const int nrows = a.B * a.T;
    const int r0 = blockIdx.z * a.rows_per_split, r1 = min(nrows, r0 + a.rows_per_split);
    for (int row = r0 + wave; row < r1; row += 4) {
        const int bb = row / a.T, t = row - bb * a.T;
        const float *gp = a.G + (((long)bb * a.Ca + (a_ok ? ai : 0)) * a.T + t) * a.Fm;
        int ts = t + toff;
        const float *sbase = a.S;
        bool s_ok = n_ok;
        if (ts < 0) { ts += a.T; sbase = a.Sprev; s_ok = s_ok && a.Sprev != nullptr && ts >= 0; }
        const float *sp = sbase ? sbase + (((long)bb * a.Cb + bch) * a.T + max(ts, 0)) * a.Fs : a.S;
        // Positions are contracted in blocks of 8: MFMA j of a block pairs positions {m0 + j, m0 + 4 + j} (the order of the contraction
        // is free), so lane half kh owns the four CONSECUTIVE positions m0 + 4 kh + [0, 4): one 16-byte load of G per block instead of four
        // scalar ones (each wave instruction gathers from 32 different rows: the instruction count was the bound, 13 TFLOP/s), and
        // the matching S window f0 + fs * [0, 4) as one or two 16-byte loads.  Windows that touch a row end take guarded scalar loads.
        struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };  // rows have odd lengths: dword-aligned 16-byte loads
        auto load_blk = [&](int m0, float (&av)[4], float (&bv)[4]) {
            const int mb = m0 + 4 * kh;
            const int f0 = a.fs * mb + kf - 2;
            if (a_ok && mb + 4 <= a.Fm) {
                const f4u q = *reinterpret_cast<const f4u *>(gp + mb);
                av[0] = q.x; av[1] = q.y; av[2] = q.z; av[3] = q.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) av[j] = (a_ok && mb + j < a.Fm) ? gp[mb + j] : 0.0f;
            }
            if (s_ok && f0 >= 0 && f0 + 3 * a.fs < a.Fs && (a.fs == 1 || f0 + 8 <= a.Fs)) {
                const f4u q0 = *reinterpret_cast<const f4u *>(sp + f0);
                if (a.fs == 1) { bv[0] = q0.x; bv[1] = q0.y; bv[2] = q0.z; bv[3] = q0.w; }
                else {
                    const f4u q1 = *reinterpret_cast<const f4u *>(sp + f0 + 4);
                    asm volatile("" ::"v"(q0.y), "v"(q0.w), "v"(q1.y), "v"(q1.w));  // keep the unused lanes "used": hipcc otherwise narrows the two
                                                                                     // 16-byte loads to four scalar ones (the instruction count is the bound)
                    bv[0] = q0.x; bv[1] = q0.z; bv[2] = q1.x; bv[3] = q1.z;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int f = f0 + a.fs * j;
                    bv[j] = (s_ok && f >= 0 && f < a.Fs) ? sp[f] : 0.0f;
                }
            }
        };
        float av[4], bv[4], an[4], bn[4];
        load_blk(0, an, bn);
        for (int m0 = 0; m0 < a.Fm; m0 += 8) {
#pragma unroll
            for (int j = 0; j < 4; j++) { av[j] = an[j]; bv[j] = bn[j]; }
            if (m0 + 8 < a.Fm) load_blk(m0 + 8, an, bn);  // one block ahead of the MFMAs
#pragma unroll
            for (int j = 0; j < 4; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
        }
    }
    // D layout: column (n) on the lane, rows (a) = (r & 3) + 8 (r >> 2) + 4 kh in the 16 registers
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0 && n_ok) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int arow = a0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (arow < a.Ca) {
                const float v = acc[r] + red[0][r][lane] + red[1][r][lane] + red[2][r][lane];
                float *dst = a.C + ((long)arow * a.Cb + bch) * a.ntap + tap;
                if (a.split_stride) dst[(long)blockIdx.z * a.split_stride] = v;
                else atomicAdd(dst, v);
            }
        }
    }
}

// C[M][N] = act(A[M][K] W[N][K]^T + bias[N]) for the training step's GEMMs, which are skinny (M = utterances x frames =
// tens to hundreds of rows, or one of N / K small): one workgroup = one 32 x 32 tile of C, its four waves split K and are
// summed through LDS, so even M = 4 (one GRU backward step) spreads over N / 32 workgroups x 4 waves instead of the
// N / 128 workgroups of the 128 x 128-tile k_gemm_tn.  fp32-exact (v_mfma_f32_32x32x2_f32); within an 8-deep k block lane
// half h contracts k = 4 h .. 4 h + 3 (one 16-byte load per operand per block) - the order of k is free as long as A and W agree.
struct SkinnyArgs {
    const float *A, *W, *bias;
    float *C;
    int M, N, K, act;
};

__global__ __launch_bounds__(256) void k_gemm_skinny(SkinnyArgs a) {
    __shared__ float red[3][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int nkb = a.K >> 3;                                  // 8-deep k blocks (K % 8 == 0: host pads)
    const int kb0 = (int)((long)nkb * wave / 4), kb1 = (int)((long)nkb * (wave + 1) / 4);
    const float *ap = a.A + (long)min(m0 + l31, a.M - 1) * a.K + 4 * kh;
    const float *wp = a.W + (long)min(n0 + l31, a.N - 1) * a.K + 4 * kh;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    constexpr int PF = 4;
    float4 qa[PF], qw[PF];
#pragma unroll
    for (int i = 0; i < PF; i++) {
        const int kb = min(kb0 + i, nkb - 1);
        qa[i] = *reinterpret_cast<const float4 *>(ap + kb * 8);
        qw[i] = *reinterpret_cast<const float4 *>(wp + kb * 8);
    }
    for (int kb = kb0; kb < kb1; kb += PF) {
#pragma unroll
        for (int i = 0; i < PF; i++) {
            const float4 ca = qa[i], cw = qw[i];
            const float s = kb + i < kb1 ? 1.0f : 0.0f;  // uniform: a slot past the wave's range contributes nothing
            const int kn = min(kb + i + PF, nkb - 1);
            qa[i] = *reinterpret_cast<const float4 *>(ap + kn * 8);
            qw[i] = *reinterpret_cast<const float4 *>(wp + kn * 8);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca.x * s, cw.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca.y * s, cw.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca.z * s, cw.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca.w * s, cw.w, acc, 0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {  // D layout: column (n) on the lane, rows (m) = (r & 3) + 8 (r >> 2) + 4 kh
        const int n = n0 + l31;
        if (n < a.N) {
            const float bs = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m < a.M) {
                    float v = acc[r] + red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + bs;
                    if (a.act == 1) v = fmaxf(v, 0.0f);
                    else if (a.act == 2) v = v > 0.0f ? v : expf(v) - 1.0f;
                    a.C[(long)m * a.N + n] = v;
                }
            }
        }
    }
}

// Gate derivatives of one GRU step (torch.nn.GRU cell; forward in gemm.hip.h: k_gru_step with `gates` saved):
//   h = (1 - z) n + z hp ;  n = tanh(gi_n + r ghn) ;  r, z = sigmoid(gi + gh)
// dh = d1 + d2 + d3 (any of them may be null): the loss gradient of this step's output plus what flows back from step t+1
// (z_{t+1} dh_{t+1}, written here as `dhz`, and dgh_{t+1} W_hh from the GEMM that follows this kernel).
struct GruBwdArgs {
    const float *d1, *d2, *d3;  // [B][H] addends of dh_t (d1 with row stride d1_ld)
    long d1_ld;
    const float *gates;         // saved r, z, n, ghn of this step: [B] rows of 4H, row stride gates_ld
    long gates_ld;
    const float *hprev;         // [B][H], row stride hprev_ld
    long hprev_ld;
    float *dgi, *dgh;           // [B] rows of 3H, row strides dg_ld
    long dg_ld;
    float *dhz;                 // [B][H]: z * dh (the direct path into h_{t-1})
    int B, H;
};

__global__ __launch_bounds__(256) void k_gru_bwd_gates(GruBwdArgs a) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)a.B * a.H) return;
    const int b = (int)(i / a.H), j = (int)(i - (long)b * a.H);
    float dh = 0.0f;
    if (a.d1) dh += a.d1[(long)b * a.d1_ld + j];
    if (a.d2) dh += a.d2[i];
    if (a.d3) dh += a.d3[i];
    const float *g = a.gates + (long)b * a.gates_ld;
    const float r = g[j], z = g[a.H + j], n = g[2 * a.H + j], ghn = g[3 * a.H + j];
    const float hp = a.hprev[(long)b * a.hprev_ld + j];
    const float dn = dh * (1.0f - z), dz = dh * (hp - n);
    const float da = dn * (1.0f - n * n);
    const float dzp = dz * z * (1.0f - z);
    const float drp = da * ghn * r * (1.0f - r);
    float *gi = a.dgi + (long)b * a.dg_ld, *gh = a.dgh + (long)b * a.dg_ld;
    gi[j] = drp; gi[a.H + j] = dzp; gi[2 * a.H + j] = da;
    gh[j] = drp; gh[a.H + j] = dzp; gh[2 * a.H + j] = da * r;
    a.dhz[i] = dh * z;
}

// C[Na][Nb] = sum_r A[r][i] B[r][j]: the weight gradient of a dense layer / 1x1 convolution, contracted over the R = streams x
// positions rows of two ROW-major operands (no transposed copies of 20-MB activations; R up to 10^6, Na, Nb <= 128).  The rows
// are split over blockIdx.z and the four waves; partial tiles are summed through LDS and added atomically into the zeroed C.
struct GemmTnArgs {
    const float *A, *B;
    float *C;
    long R, rows_per_split;
    int Na, Nb;
    long split_stride;  // > 0: split z writes its partial tile to C + z * split_stride (no atomics); 0: atomicAdd into C
};

__global__ __launch_bounds__(256) void k_gemm_tn_acc(GemmTnArgs a) {
    __shared__ float red[3][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const long r0 = (long)blockIdx.z * a.rows_per_split, r1 = min(a.R, r0 + a.rows_per_split);
    const int ia = min(i0 + l31, a.Na - 1), jb = min(j0 + l31, a.Nb - 1);
    const bool ia_ok = i0 + l31 < a.Na, jb_ok = j0 + l31 < a.Nb;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    constexpr int U = 8;  // MFMA steps per trip: row r0 + 8 (U s + u) + 2 wave + kh
    for (long base = r0 + 2 * wave + kh; base < r1; base += 8 * U) {
        float av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long row = base + 8 * u;
            const bool ok = row < r1;
            const long rc = ok ? row : r1 - 1;
            av[u] = (ok && ia_ok) ? a.A[rc * a.Na + ia] : 0.0f;
            bv[u] = (ok && jb_ok) ? a.B[rc * a.Nb + jb] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0 && jb_ok) {  // D layout: column (j) on the lane, rows (i) = (r & 3) + 8 (r >> 2) + 4 kh
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (i < a.Na) {
                const float v = acc[r] + red[0][r][lane] + red[1][r][lane] + red[2][r][lane];
                float *dst = a.C + (long)i * a.Nb + j0 + l31;
                if (a.split_stride) dst[(long)blockIdx.z * a.split_stride] = v;
                else atomicAdd(dst, v);
            }
        }
    }
}

// One BPTT step in ONE launch: the gate derivatives of step t for every (stream, unit) and the hand-over to step t - 1,
//   g_t[b][k] = sum_j dgh_t[b][j] W_hh[j][k]      (w_hh_t = W_hh^T, [H][3H], K-contiguous like k_gemm_skinny's operands)
// Every workgroup recomputes the (cheap, B x H element) gate derivatives into LDS, the owner of a 32-unit column block also
// writes them out, then contracts its block of W_hh^T against the LDS copy (32x32x2 fp32 MFMA, K = 3H split over the four
// waves, 12 eight-deep k blocks in flight per wave).  Replaces k_gru_bwd_gates + k_gemm_skinny (5 + 25 us, two dependent
// launches per step) for micro-batches of up to 16 streams.
struct GruBwdStepArgs {
    GruBwdArgs g;        // as k_gru_bwd_gates
    const float *whh_t;  // [H][3H]
    float *gout;         // [B][H] = dgh_t W_hh, or null for the first step of the sequence (nobody consumes it)
};
constexpr int kGruBwdMaxB = 16;

constexpr int kGruBwdWaves = 8;  // K = 3H split over eight waves: 24 eight-deep k blocks per wave at H = 512 = two rounds of 12 in flight

__global__ __launch_bounds__(kGruBwdWaves * 64) void k_gru_bwd_step(GruBwdStepArgs s) {
    extern __shared__ __align__(16) float dl[];  // dgh_t [B][3H + 4]
    __shared__ float red[kGruBwdWaves - 1][16][64];
    const GruBwdArgs &a = s.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = a.H, K = 3 * H, ldl = K + 4;
    const int n0 = blockIdx.x * 32;
    // four elements per thread per trip, every load of the trip issued before the first use (the trips are a chain of L2 round
    // trips otherwise: the stores of one element fence the loads of the next)
    const int total = a.B * H;
    constexpr int NTH = kGruBwdWaves * 64;
    for (int i0 = tid; i0 < total; i0 += 4 * NTH) {
        float v1[4], v2[4], v3[4], gr[4], gz[4], gn[4], gg[4], hv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = min(i0 + NTH * u, total - 1), b = i / H, j = i - b * H;
            const float *g = a.gates + (long)b * a.gates_ld;
            v1[u] = a.d1 ? a.d1[(long)b * a.d1_ld + j] : 0.0f;
            v2[u] = a.d2 ? a.d2[i] : 0.0f;
            v3[u] = a.d3 ? a.d3[i] : 0.0f;
            gr[u] = g[j]; gz[u] = g[H + j]; gn[u] = g[2 * H + j]; gg[u] = g[3 * H + j];
            hv[u] = a.hprev[(long)b * a.hprev_ld + j];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + NTH * u;
            if (i >= total) break;
            const int b = i / H, j = i - b * H;
            const float dh = (v1[u] + v2[u]) + v3[u];
            const float r = gr[u], z = gz[u], n = gn[u], ghn = gg[u], hp = hv[u];
            const float dn = dh * (1.0f - z), dz = dh * (hp - n);
            const float da = dn * (1.0f - n * n);
            const float dzp = dz * z * (1.0f - z);
            const float drp = da * ghn * r * (1.0f - r);
            float *l = dl + b * ldl;
            l[j] = drp; l[H + j] = dzp; l[2 * H + j] = da * r;
            if (j >= n0 && j < n0 + 32) {  // this workgroup's units: the step's outputs
                float *gi = a.dgi + (long)b * a.dg_ld, *gh = a.dgh + (long)b * a.dg_ld;
                gi[j] = drp; gi[H + j] = dzp; gi[2 * H + j] = da;
                gh[j] = drp; gh[H + j] = dzp; gh[2 * H + j] = da * r;
                a.dhz[i] = dh * z;
            }
        }
    }
    if (!s.gout) return;
    __syncthreads();
    const int l31 = lane & 31, kh = lane >> 5;
    const int nkb = K >> 3;  // 3H is a multiple of 48
    const int kb0 = (int)((long)nkb * wave / kGruBwdWaves), kb1 = (int)((long)nkb * (wave + 1) / kGruBwdWaves);
    const float *ap = dl + min(l31, a.B - 1) * ldl + 4 * kh;
    const float *wp = s.whh_t + (long)min(n0 + l31, H - 1) * K + 4 * kh;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    constexpr int PF = 12;
    float4 qw[PF];
#pragma unroll
    for (int i = 0; i < PF; i++) qw[i] = *reinterpret_cast<const float4 *>(wp + min(kb0 + i, nkb - 1) * 8);
    for (int kb = kb0; kb < kb1; kb += PF) {
#pragma unroll
        for (int i = 0; i < PF; i++) {
            const float4 cw = qw[i];
            const float sc = kb + i < kb1 ? 1.0f : 0.0f;  // uniform: a slot past the wave's range contributes nothing
            const float4 ca = *reinterpret_cast<const float4 *>(ap + min(kb + i, nkb - 1) * 8);
            qw[i] = *reinterpret_cast<const float4 *>(wp + min(kb + i + PF, nkb - 1) * 8);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca.x * sc, cw.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca.y * sc, cw.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca.z * sc, cw.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca.w * sc, cw.w, acc, 0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0 && n0 + l31 < H) {  // D layout: column (unit) on the lane, rows (stream) = (r & 3) + 8 (r >> 2) + 4 kh
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (m < a.B) {
                float v = acc[r];
#pragma unroll
                for (int w = 0; w < kGruBwdWaves - 1; w++) v += red[w][r][lane];
                s.gout[(long)m * H + n0 + l31] = v;
            }
        }
    }
}

}  // namespace se

extern "C" {

int se_train_conv_layout_query(int kind, int Ci, int Co, int T, int Fi, int Fy, int dil, se_train_conv_layout *out) {
    if (!out) return tfail(SE_ERR_ARG, "null argument");
    TrainConvGeo g;
    int rc = train_conv_geometry(kind, Ci, Co, T, Fi, Fy, dil, g);
    if (rc) return rc;
    out->ntap = g.a.ntap; out->CC = g.a.CC; out->nchunk = g.a.nchunk; out->CoPad = g.a.CoPad; out->FP = g.a.FP;
    for (int t = 0; t < 15; t++) { out->tap_kf[t] = t < g.a.ntap ? g.tap_kf[t] : 0; out->tap_kt[t] = t < g.a.ntap ? g.tap_kt[t] : 0; }
    return SE_OK;
}

int se_train_conv(int kind, const float *x, const float *xprev, const float *w_arranged, const float *bias, float *y, int B, int Ci, int Co,
                  int T, int Fi, int Fy, int dil, int act, void *stream) {
    if (!x || !w_arranged || !bias || !y || B <= 0) return tfail(SE_ERR_ARG, "null argument");
    TrainConvGeo g;
    int rc = train_conv_geometry(kind, Ci, Co, T, Fi, Fy, dil, g);
    if (rc) return rc;
    ConvArgs a = g.a;
    a.x = x; a.xprev = xprev; a.w = w_arranged; a.bias = bias; a.y = y;
    a.act = act; a.relu_lo = 0; a.relu_hi = act ? Co : 0;
    train_conv_attributes();
    if (conv_igemm_launch(a.ntap, g.NT, a.CoPad, dim3(g.grid_x, B), g.lds, static_cast<hipStream_t>(stream), a))
        return tfail(SE_ERR_ARG, "no conv kernel instance for %d taps x %d tiles", a.ntap, g.NT);
    return hipGetLastError() == hipSuccess ? SE_OK : tfail(SE_ERR_HIP, "conv launch failed");
}

int se_train_conv_wgrad(const float *G, const float *S, const float *Sprev, float *C, int B, int Ca, int Cb, int T, int Fm, int Fs, int dil,
                        void *stream) {
    if (!G || !S || !C || B <= 0) return tfail(SE_ERR_ARG, "null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(C, 0, (size_t)Ca * Cb * 15 * sizeof(float), st) != hipSuccess) return tfail(SE_ERR_HIP, "memset failed");
    const int nrows = B * T, tiles = ((Cb * 15 + 31) / 32) * ((Ca + 31) / 32);
    int nsplit = std::max(1, std::min(nrows / 4, (4 * 256 + tiles - 1) / tiles));  // ~4 workgroups per CU over the whole grid
    const int rows_per_split = (nrows + nsplit - 1) / nsplit;
    nsplit = (nrows + rows_per_split - 1) / rows_per_split;
    se::WgradArgs a{G, S, Sprev, C, B, Ca, Cb, T, Fm, Fs, dil, rows_per_split, 15, 2, 0};
    hipLaunchKernelGGL(se::k_corr_wgrad, dim3((Cb * 15 + 31) / 32, (Ca + 31) / 32, nsplit), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? SE_OK : tfail(SE_ERR_HIP, "wgrad launch failed");
}

/* Deterministic weight gradient (no atomics): the (batch, t) rows are split over `nsplit` workgroup layers which write partial
 * tiles to ws[nsplit][Ca*Cb*ntap]; se_train_colsum folds them in a fixed order.  ntap = 15 (5x3 kernels, frequency stride 2) or
 * 1 (1x1 convolutions).  Returns the number of splits through *nsplit_out; ws needs se_train_wgrad_ws_floats() floats. */
int se_train_conv_wgrad_det(const float *G, const float *S, const float *Sprev, float *ws, int *nsplit_out, int B, int Ca, int Cb, int T, int Fm,
                            int Fs, int dil, int ntap, void *stream) {
    if (!G || !S || !ws || !nsplit_out || B <= 0 || (ntap != 15 && ntap != 1)) return tfail(SE_ERR_ARG, "bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nrows = B * T, ncol = Cb * ntap, tiles = ((ncol + 31) / 32) * ((Ca + 31) / 32);
    int nsplit = std::max(1, std::min(std::min(nrows / 4, 64), (4 * 256 + tiles - 1) / tiles));
    const int rows_per_split = (nrows + nsplit - 1) / nsplit;
    nsplit = (nrows + rows_per_split - 1) / rows_per_split;
    se::WgradArgs a{G, S, Sprev, ws, B, Ca, Cb, T, Fm, Fs, dil, rows_per_split, ntap, ntap == 1 ? 1 : 2, (long)Ca * Cb * ntap};
    hipLaunchKernelGGL(se::k_corr_wgrad, dim3((ncol + 31) / 32, (Ca + 31) / 32, nsplit), dim3(256), 0, st, a);
    *nsplit_out = nsplit;
    return hipGetLastError() == hipSuccess ? SE_OK : tfail(SE_ERR_HIP, "wgrad launch failed");
}

/* se_train_conv with the weights in their checkpoint layout: element (co, ci, kf, kt) of the GEMM-row-major view at
 * w[co * sCo + ci * sCi + kf * 3 + kt] (Conv2d: sCo = Ci*15, sCi = 15; ConvTranspose2d [Cin][Cout][5][3] read with rows = its Cout:
 * sCo = 15, sCi = Cout*15; 1x1: sCo = Ci, sCi = 1 or transposed).  The arrangement the kernel stages is made on the device into `ws`
 * (se_train_conv_ws_floats() floats) by one small launch ahead of the convolution: no host-side tensor shuffling per step. */
int se_train_conv_ws_floats(int kind, int Ci, int Co, int T, int Fi, int Fy, int dil) {
    TrainConvGeo g;
    if (train_conv_geometry(kind, Ci, Co, T, Fi, Fy, dil, g)) return -1;
    return g.a.nchunk * g.a.ntap * g.a.CC * g.a.CoPad;
}
int se_train_conv_w(int kind, const float *x, const float *xprev, const float *w, int64_t sCo, int64_t sCi, const float *bias, float *y, float *ws,
                    int B, int Ci, int Co, int T, int Fi, int Fy, int dil, int act, int Cy, int cy0, void *stream) {
    if (!x || !w || !bias || !y || !ws || B <= 0) return tfail(SE_ERR_ARG, "null argument");
    TrainConvGeo g;
    int rc = train_conv_geometry(kind, Ci, Co, T, Fi, Fy, dil, g);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    se::launch_arrange_w(w, ws, (long)sCo, (long)sCi, Co, Ci, g.a.ntap, g.a.CC, g.a.nchunk, g.a.CoPad, kind == 3, g.tap_kf, g.tap_kt, st);
    ConvArgs a = g.a;
    a.x = x; a.xprev = xprev; a.w = ws; a.bias = bias; a.y = y;
    a.act = act; a.relu_lo = 0; a.relu_hi = act ? Co : 0;
    if (Cy > 0) {  // the Co output channels land at channels [cy0, cy0 + Co) of a tensor with Cy channels per stream
        if (cy0 < 0 || cy0 + Co > Cy) return tfail(SE_ERR_ARG, "channel window [%d, %d) outside %d channels", cy0, cy0 + Co, Cy);
        a.Cy = Cy; a.cy0 = cy0;
    }
    train_conv_attributes();
    if (conv_igemm_launch(a.ntap, g.NT, a.CoPad, dim3(g.grid_x, B), g.lds, st, a))
        return tfail(SE_ERR_ARG, "no conv kernel instance for %d taps x %d tiles", a.ntap, g.NT);
    return hipGetLastError() == hipSuccess ? SE_OK : tfail(SE_ERR_HIP, "conv launch failed");
}

int se_train_gemm(const float *A, const float *W, const float *bias, float *C, int M, int N, int K, int act, void *stream) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return tfail(SE_ERR_ARG, "null argument");
    if (K % 8) return tfail(SE_ERR_ARG, "GEMM inner dimension %d must be a multiple of 8 (pad with zeros)", K);
    se::SkinnyArgs g{A, W, bias, C, M, N, K, act};
    hipLaunchKernelGGL(se::k_gemm_skinny, dim3((N + 31) / 32, (M + 31) / 32), dim3(256), 0, static_cast<hipStream_t>(stream), g);
    return hipGetLastError() == hipSuccess ? SE_OK : tfail(SE_ERR_HIP, "gemm launch failed");
}

/* C[Na][Nb] = sum over the R rows of A[r][i] * B[r][j] (both row-major): weight gradients without transposed copies */
int se_train_gemm_tn(const float *A, const float *B, float *C, int64_t R, int Na, int Nb, void *stream) {
    if (!A || !B || !C || R <= 0 || Na <= 0 || Nb <= 0) return tfail(SE_ERR_ARG, "null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(C, 0, (size_t)Na * Nb * sizeof(float), st) != hipSuccess) return tfail(SE_ERR_HIP, "memset failed");
    const int tiles = ((Na + 31) / 32) * ((Nb + 31) / 32);
    long nsplit = std::max<long>(1, std::min<long>((R + 255) / 256, (4 * 256 + tiles - 1) / tiles));  // ~4 workgroups per CU, >= 256 rows each
    const long rows_per_split = ((R + nsplit - 1) / nsplit + 7) / 8 * 8;
    nsplit = (R + rows_per_split - 1) / rows_per_split;
    se::GemmTnArgs a{A, B, C, (long)R, rows_per_split, Na, Nb, 0};
    hipLaunchKernelGGL(se::k_gemm_tn_acc, dim3((Nb + 31) / 32, (Na + 31) / 32, (unsigned)nsplit), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? SE_OK : tfail(SE_ERR_HIP, "gemm_tn launch failed");
}

/* deterministic form: partial tiles per row split to ws[nsplit][Na*Nb] (<= 64 splits), folded by se_train_colsum */
int se_train_gemm_tn_det(const float *A, const float *B, float *ws, int *nsplit_out, int64_t R, int Na, int Nb, void *stream) {
    if (!A || !B || !ws || !nsplit_out || R <= 0 || Na <= 0 || Nb <= 0) return tfail(SE_ERR_ARG, "null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int tiles = ((Na + 31) / 32) * ((Nb + 31) / 32);
    long nsplit = std::max<long>(1, std::min<long>(std::min<long>((R + 255) / 256, 64), (4 * 256 + tiles - 1) / tiles));
    const long rows_per_split = ((R + nsplit - 1) / nsplit + 7) / 8 * 8;
    nsplit = (R + rows_per_split - 1) / rows_per_split;
    se::GemmTnArgs a{A, B, ws, (long)R, rows_per_split, Na, Nb, (long)Na * Nb};
    hipLaunchKernelGGL(se::k_gemm_tn_acc, dim3((Nb + 31) / 32, (Na + 31) / 32, (unsigned)nsplit), dim3(256), 0, st, a);
    *nsplit_out = (int)nsplit;
    return hipGetLastError() == hipSuccess ? SE_OK : tfail(SE_ERR_HIP, "gemm_tn launch failed");
}

int se_train_gru_step(const float *gi, int64_t gi_ld, const float *hprev, const float *whh, const float *bhh, float *hout, float *seq,
                      int64_t seq_ld, float *gates, int64_t gates_ld, int B, int H, void *stream) {
    if (!gi || !hprev || !whh || !bhh || !hout || !seq || B <= 0 || H <= 0 || H % 16) return tfail(SE_ERR_ARG, "bad argument (H must be a multiple of 16)");
    se::GruStepArgs g{gi, (long)gi_ld, hprev, whh, bhh, hout, seq, (long)seq_ld, B, H, gates, (long)gates_ld};
    hipLaunchKernelGGL(se::k_gru_step, dim3((H + 15) / 16, (B + 31) / 32), dim3(256), 0, static_cast<hipStream_t>(stream), g);
    return hipGetLastError() == hipSuccess ? SE_OK : tfail(SE_ERR_HIP, "gru step launch failed");
}

int se_train_gru_bwd_gates(const float *d1, int64_t d1_ld, const float *d2, const float *d3, const float *gates, int64_t gates_ld, const float *hprev,
                           int64_t hprev_ld, float *dgi, float *dgh, int64_t dg_ld, float *dhz, int B, int H, void *stream) {
    if (!gates || !hprev || !dgi || !dgh || !dhz || B <= 0 || H <= 0) return tfail(SE_ERR_ARG, "null argument");
    se::GruBwdArgs a{d1, d2, d3, (long)d1_ld, gates, (long)gates_ld, hprev, (long)hprev_ld, dgi, dgh, (long)dg_ld, dhz, B, H};
    hipLaunchKernelGGL(se::k_gru_bwd_gates, dim3((unsigned)(((long)B * H + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SE_OK : tfail(SE_ERR_HIP, "gru backward launch failed");
}

/* All T steps of one GRU layer from C (one Python call instead of T): forward with the gate values saved, and the BPTT
 * sweep with truncation at segment boundaries (the carried state is detached at every segment, CRN.py:281): the
 * gradient is not carried from step t + 1 into step t when (t + 1) % seg_len == 0.  scratch: forward 2 * B * H floats,
 * backward 4 * B * H floats. */
int se_train_gru_seq_fwd(const float *gi, const float *h0, const float *whh, const float *bhh, float *out, float *gates, float *hT, float *scratch,
                         int B, int T, int H, void *stream) {
    if (!gi || !h0 || !whh || !bhh || !out || !gates || !hT || !scratch || B <= 0 || T <= 0 || H <= 0 || H % 16) return tfail(SE_ERR_ARG, "bad argument (H must be a multiple of 16)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *hb[2] = {scratch, scratch + (size_t)B * H};
    if (hipMemcpyAsync(hb[0], h0, (size_t)B * H * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return tfail(SE_ERR_HIP, "copy failed");
    for (int t = 0; t < T; t++) {
        se::GruStepArgs g{gi + (size_t)t * 3 * H, (long)T * 3 * H, hb[t & 1], whh, bhh, hb[(t + 1) & 1], out + (size_t)t * H, (long)T * H, B, H,
                          gates + (size_t)t * 4 * H, (long)T * 4 * H};
        if (B <= 16) hipLaunchKernelGGL(se::k_gru_step8, dim3((H + 15) / 16, 1), dim3(512), 0, st, g);
        else hipLaunchKernelGGL(se::k_gru_step, dim3((H + 15) / 16, (B + 31) / 32), dim3(256), 0, st, g);
    }
    if (hipMemcpyAsync(hT, hb[T & 1], (size_t)B * H * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return tfail(SE_ERR_HIP, "copy failed");
    return hipGetLastError() == hipSuccess ? SE_OK : tfail(SE_ERR_HIP, "gru sequence launch failed");
}

int se_train_gru_seq_bwd(const float *dout, const float *dhT, const float *gates, const float *out, const float *h0, const float *whh_t, float *dgi,
                         float *dgh, float *scratch, int B, int T, int H, int seg_len, void *stream) {
    if (!dout || !gates || !out || !h0 || !whh_t || !dgi || !dgh || !scratch || B <= 0 || T <= 0 || H <= 0 || H % 16) return tfail(SE_ERR_ARG, "bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *zb[2] = {scratch, scratch + (size_t)B * H}, *gb[2] = {scratch + (size_t)2 * B * H, scratch + (size_t)3 * B * H};
    if (B > se::kGruBwdMaxB) return tfail(SE_ERR_ARG, "the fused BPTT step takes up to %d streams per call (got %d): use the per-step entry points", se::kGruBwdMaxB, B);
    const size_t lds = (size_t)B * (3 * H + 4) * sizeof(float);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(se::k_gru_bwd_step), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); attr = true; }
    if (lds > 128 * 1024) return tfail(SE_ERR_ARG, "hidden size %d too large for the fused BPTT step", H);
    for (int t = T - 1; t >= 0; t--) {
        const bool last = t == T - 1, cut = !last && seg_len > 0 && (t + 1) % seg_len == 0;
        const float *d2 = last ? dhT : (cut ? nullptr : zb[(t + 1) & 1]);
        const float *d3 = (last || cut) ? nullptr : gb[(t + 1) & 1];
        const float *hp = t == 0 ? h0 : out + (size_t)(t - 1) * H;
        const long hp_ld = t == 0 ? H : (long)T * H;
        se::GruBwdArgs a{dout + (size_t)t * H, d2, d3, (long)T * H, gates + (size_t)t * 4 * H, (long)T * 4 * H, hp, hp_ld,
                         dgi + (size_t)t * 3 * H, dgh + (size_t)t * 3 * H, (long)T * 3 * H, zb[t & 1], B, H};
        se::GruBwdStepArgs s{a, whh_t, t > 0 ? gb[t & 1] : nullptr};
        hipLaunchKernelGGL(se::k_gru_bwd_step, dim3((H + 31) / 32), dim3(se::kGruBwdWaves * 64), lds, st, s);
    }
    return hipGetLastError() == hipSuccess ? SE_OK : tfail(SE_ERR_HIP, "gru backward sequence launch failed");
}

}  // extern "C"
