// train_fused.hip.h - the norm / pointwise / signal stages of the TRAINING step as hand-written kernels, forward AND
// backward (SURVEY.md 8f-1 remainder; reference = torch autograd over CRN.py:111-159 GlobalLayerNorm, 387-396 skip gate,
// 463-467 features, 491-495 mask application, utility.py:439-442 decompress_cIRM, 373-403 over_add).
//
// Activations are [S][C][T][F] fp32 (F innermost; S = N segments x B utterances, segment-major).  Every norm is per stream:
// one workgroup owns one stream (<= 45 k elements, L2-resident after the first pass) and walks it in passes; per-channel
// parameter-gradient partials go to [S][C] slabs (deterministic: one slot per stream, fixed summation order), which
// k_colsum folds over S.  No atomics anywhere.
//
// GlobalLayerNorm (CRN.py:135-149): y = (a - mean) * inv * w + b, a = act(x), inv = 1 / (sqrt(var + eps) + eps), biased var.
// Backward with g = dy * w, xh = (a - mean) * inv, sd = sqrt(var + eps) = 1/inv - eps:
//   da = inv * (g - mean(g)) - xh * mean(g * xh) / sd        (d inv / d var = -inv^2 / (2 sd): not the textbook -inv^3 / 2)
//   dx = da * act'(x);   dw = sum dy * xh;   db = sum dy;   the producer's bias gradient = sum dx per channel.
#pragma once
#include <hip/hip_runtime.h>
#include "fft_lds.h"

namespace se {

constexpr float kTEps = 1e-8f;  // CRN.py:11

__device__ __forceinline__ float t_act(float v, int act) { return act == 1 ? fmaxf(v, 0.0f) : (act == 2 ? (v > 0.0f ? v : expf(v) - 1.0f) : v); }
__device__ __forceinline__ float t_dact(float v, int act) { return act == 1 ? (v > 0.0f ? 1.0f : 0.0f) : (act == 2 ? (v > 0.0f ? 1.0f : expf(v)) : 1.0f); }

// The per-stream kernels run kTW waves per workgroup: a stream has up to 45 k elements and is walked in 2-3 dependent passes, so the
// pass length (elements / threads) is the launch time; 16 waves instead of 4: k_tgln_fwd 194 -> 107 us, k_tgln_bwd 218 -> 82 us per launch
// (272 streams, measured in bench.py --mode train).
constexpr int kTW = 16, kTT = kTW * 64;

// block-wide sums of up to two values over kTT threads, combined in double in a fixed order; every thread gets the result
__device__ __forceinline__ void t_block_sum2(double &a, double &b, double *red /*[2 * kTW]*/) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { red[wave * 2] = a; red[wave * 2 + 1] = b; }
    __syncthreads();
    double sa = 0, sb = 0;
#pragma unroll
    for (int w = 0; w < kTW; w++) { sa += red[2 * w]; sb += red[2 * w + 1]; }
    a = sa; b = sb;
}

__device__ __forceinline__ float t_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;  // valid on lane 0
}

struct TGlnArgs {
    const float *x;       // pre-activation; element (s, c, t, f) at s * xS + c * xC + t * xT + f
    long xS, xC, xT;
    float *y;             // forward out / backward dx (dx uses the x strides)
    long yS, yC, yT;
    const float *dy;      // backward: upstream gradient, element at s * dS + c * dC + t * dT + f (only f < Fi is read)
    long dS, dC, dT;
    const float *w, *b;   // affine, index c (mode 0) or c * Fi + f (mode 1 = GlobalLayerNorm(last=True), CRN.py:127-129)
    float *stats;         // [S][2] = mean, inv (forward writes, backward reads)
    float *dw_part, *db_part, *dpre_part;  // backward: [S][NA] slabs, NA = C (mode 0) or C * Fi (mode 1)
    int C, T, Fi, Fo;     // Fo >= Fi: the forward zero-fills columns [Fi, Fo) (the decoder's frequency pad, CRN.py:389-392)
    int mode, act, eps_mode;
};

// forward: y = gLN(act(x)); one workgroup per stream, three passes (mean, variance, write)
__global__ __launch_bounds__(kTT) void k_tgln_fwd(TGlnArgs a) {
    __shared__ double red[2 * kTW];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int TF = a.T * a.Fi, n = a.C * TF;
    const float *x = a.x + (long)s * a.xS;
    double sum = 0, dummy = 0;
    for (int e = tid; e < n; e += kTT) {
        const int c = e / TF, r = e - c * TF, t = r / a.Fi, f = r - t * a.Fi;
        sum += (double)t_act(x[c * a.xC + t * a.xT + f], a.act);
    }
    t_block_sum2(sum, dummy, red);
    const float mean = (float)(sum / n);
    double sq = 0;
    dummy = 0;
    for (int e = tid; e < n; e += kTT) {
        const int c = e / TF, r = e - c * TF, t = r / a.Fi, f = r - t * a.Fi;
        const float d = t_act(x[c * a.xC + t * a.xT + f], a.act) - mean;
        sq += (double)(d * d);
    }
    t_block_sum2(sq, dummy, red);
    const float var = (float)(sq / n);
    const float inv = 1.0f / ((a.eps_mode ? sqrtf(var) : sqrtf(var + kTEps)) + kTEps);
    if (tid == 0) { a.stats[2 * s] = mean; a.stats[2 * s + 1] = inv; }
    float *y = a.y + (long)s * a.yS;
    const int TFo = a.T * a.Fo, no = a.C * TFo;
    for (int e = tid; e < no; e += kTT) {
        const int c = e / TFo, r = e - c * TFo, t = r / a.Fo, f = r - t * a.Fo;
        float v = 0.0f;
        if (f < a.Fi) {
            const int ai = a.mode ? c * a.Fi + f : c;
            v = (t_act(x[c * a.xC + t * a.xT + f], a.act) - mean) * inv * a.w[ai] + a.b[ai];
        }
        y[c * a.yC + t * a.yT + f] = v;
    }
}

// backward, mode 0 (per-channel affine): wave w owns channels c = w, w + 4, ... so the per-channel sums need no atomics
__global__ __launch_bounds__(kTT) void k_tgln_bwd_c(TGlnArgs a) {
    __shared__ double red[2 * kTW];
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int TF = a.T * a.Fi, n = a.C * TF;
    const float *x = a.x + (long)s * a.xS;
    const float *dy = a.dy + (long)s * a.dS;
    const float mean = a.stats[2 * s], inv = a.stats[2 * s + 1];
    double S1 = 0, S2 = 0;
    for (int c = wave; c < a.C; c += kTW) {
        const float wc = a.w[c];
        float pw = 0.0f, pb = 0.0f;
        for (int r = lane; r < TF; r += 64) {
            const int t = r / a.Fi, f = r - t * a.Fi;
            const float xh = (t_act(x[c * a.xC + t * a.xT + f], a.act) - mean) * inv;
            const float d = dy[c * a.dC + t * a.dT + f];
            pw += d * xh; pb += d;
        }
        pw = t_wave_sum(pw); pb = t_wave_sum(pb);
        if (lane == 0) {
            if (a.dw_part) { a.dw_part[(long)s * a.C + c] = pw; a.db_part[(long)s * a.C + c] = pb; }
            S1 += (double)(wc * pb); S2 += (double)(wc * pw);
        }
    }
    t_block_sum2(S1, S2, red);
    const float sd = 1.0f / inv - kTEps;
    const float m1 = (float)(S1 / n), m2 = (float)(S2 / n) / (a.eps_mode ? fmaxf(sd, 1e-30f) : sd);
    float *dx = a.y + (long)s * a.xS;
    for (int c = wave; c < a.C; c += kTW) {
        const float wc = a.w[c];
        float pp = 0.0f;
        for (int r = lane; r < TF; r += 64) {
            const int t = r / a.Fi, f = r - t * a.Fi;
            const long xo = c * a.xC + t * a.xT + f;
            const float xv = x[xo];
            const float xh = (t_act(xv, a.act) - mean) * inv;
            const float g = dy[c * a.dC + t * a.dT + f] * wc;
            const float v = (inv * (g - m1) - xh * m2) * t_dact(xv, a.act);
            dx[xo] = v;
            pp += v;
        }
        pp = t_wave_sum(pp);
        if (lane == 0 && a.dpre_part) a.dpre_part[(long)s * a.C + c] = pp;
    }
}

// backward, mode 1 (per-feature affine, d = c * Fi + f; the norm after fc_output_layer): thread owns features d = tid, tid + 256, ...
__global__ __launch_bounds__(kTT) void k_tgln_bwd_d(TGlnArgs a) {
    __shared__ double red[2 * kTW];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int D = a.C * a.Fi, n = D * a.T;
    const float *x = a.x + (long)s * a.xS;
    const float *dy = a.dy + (long)s * a.dS;
    const float mean = a.stats[2 * s], inv = a.stats[2 * s + 1];
    double S1 = 0, S2 = 0;
    for (int d = tid; d < D; d += kTT) {
        const int c = d / a.Fi, f = d - c * a.Fi;
        const float wd = a.w[d];
        float pw = 0.0f, pb = 0.0f;
        for (int t = 0; t < a.T; t++) {
            const float xh = (t_act(x[c * a.xC + t * a.xT + f], a.act) - mean) * inv;
            const float g = dy[c * a.dC + t * a.dT + f];
            pw += g * xh; pb += g;
        }
        if (a.dw_part) { a.dw_part[(long)s * D + d] = pw; a.db_part[(long)s * D + d] = pb; }
        S1 += (double)(wd * pb); S2 += (double)(wd * pw);
    }
    t_block_sum2(S1, S2, red);
    const float sd = 1.0f / inv - kTEps;
    const float m1 = (float)(S1 / n), m2 = (float)(S2 / n) / (a.eps_mode ? fmaxf(sd, 1e-30f) : sd);
    float *dx = a.y + (long)s * a.xS;
    for (int d = tid; d < D; d += kTT) {
        const int c = d / a.Fi, f = d - c * a.Fi;
        const float wd = a.w[d];
        float pp = 0.0f;
        for (int t = 0; t < a.T; t++) {
            const long xo = c * a.xC + t * a.xT + f;
            const float xv = x[xo];
            const float xh = (t_act(xv, a.act) - mean) * inv;
            const float g = dy[c * a.dC + t * a.dT + f] * wd;
            const float v = (inv * (g - m1) - xh * m2) * t_dact(xv, a.act);
            dx[xo] = v;
            pp += v;
        }
        if (a.dpre_part) a.dpre_part[(long)s * D + d] = pp;
    }
}

// out[j] (+)= sum_r part[r][j] for up to three slabs of R rows (fixed order: deterministic)
struct TColsumArgs {
    const float *part[3];
    float *out[3];
    int n[3];
    int R, accumulate;
};
__global__ __launch_bounds__(256) void k_colsum(TColsumArgs a) {
    const int k = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (!a.part[k] || j >= a.n[k]) return;
    const float *p = a.part[k] + j;
    const int n = a.n[k];
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int r = 0;
    for (; r + 4 <= a.R; r += 4) { s0 += p[(long)r * n]; s1 += p[(long)(r + 1) * n]; s2 += p[(long)(r + 2) * n]; s3 += p[(long)(r + 3) * n]; }
    for (; r < a.R; r++) s0 += p[(long)r * n];
    const float v = (s0 + s1) + (s2 + s3);
    a.out[k][j] = a.accumulate ? a.out[k][j] + v : v;
}
// two-stage column sum for tall matrices (GRU bias gradients: R = streams x frames): stage 1 sums chunks of 64 rows
__global__ __launch_bounds__(256) void k_colsum_chunks(const float *x, float *part, int R, int n) {
    const int j = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y;
    if (j >= n) return;
    const int r0 = ch * 64, r1 = min(R, r0 + 64);
    float s = 0;
    for (int r = r0; r < r1; r++) s += x[(long)r * n + j];
    part[(long)ch * n + j] = s;
}

// ---- decoder skip gate (CRN.py:387-396): out = m * act(u) + (1 - m) * z, m = sigmoid(gLN(v)) ------------------------------
// uv [S][2 Co][T][F]: channels [0, Co) = residual(res) = u, [Co, 2 Co) = residualmask(res) = v;  z [S][Co][T][F] = padded gLN(act(deconv))
struct TSkipArgs {
    const float *uv, *z, *dout;
    float *out;            // forward
    float *duv, *dz;       // backward
    const float *nw, *nb;  // residualnorm affine [Co]
    float *stats;          // [S][2] of v
    float *dnw_part, *dnb_part, *dbias_part;  // [S][Co], [S][Co], [S][2 Co] (bias gradients of the stacked 1x1 convolution)
    int Co, T, F, act, eps_mode;
};

__global__ __launch_bounds__(kTT) void k_tskip_fwd(TSkipArgs a) {
    __shared__ double red[2 * kTW];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int TF = a.T * a.F, n = a.Co * TF;
    const float *u = a.uv + (long)s * 2 * n, *v = u + n;
    double sum = 0, dummy = 0;
    for (int e = tid; e < n; e += kTT) sum += (double)v[e];
    t_block_sum2(sum, dummy, red);
    const float mean = (float)(sum / n);
    double sq = 0;
    dummy = 0;
    for (int e = tid; e < n; e += kTT) { const float d = v[e] - mean; sq += (double)(d * d); }
    t_block_sum2(sq, dummy, red);
    const float var = (float)(sq / n);
    const float inv = 1.0f / ((a.eps_mode ? sqrtf(var) : sqrtf(var + kTEps)) + kTEps);
    if (tid == 0) { a.stats[2 * s] = mean; a.stats[2 * s + 1] = inv; }
    const float *z = a.z + (long)s * n;
    float *o = a.out + (long)s * n;
    for (int e = tid; e < n; e += kTT) {
        const int c = e / TF;
        const float vn = (v[e] - mean) * inv * a.nw[c] + a.nb[c];
        const float m = 1.0f / (1.0f + expf(-vn));
        o[e] = m * t_act(u[e], a.act) + (1.0f - m) * z[e];
    }
}

__global__ __launch_bounds__(kTT) void k_tskip_bwd(TSkipArgs a) {
    __shared__ double red[2 * kTW];
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int TF = a.T * a.F, n = a.Co * TF;
    const float *u = a.uv + (long)s * 2 * n, *v = u + n;
    const float *z = a.z + (long)s * n, *dout = a.dout + (long)s * n;
    float *du = a.duv + (long)s * 2 * n, *dv = du + n, *dz = a.dz + (long)s * n;
    const float mean = a.stats[2 * s], inv = a.stats[2 * s + 1];
    double S1 = 0, S2 = 0;
    for (int c = wave; c < a.Co; c += kTW) {
        const float wc = a.nw[c], bc = a.nb[c];
        float pw = 0.0f, pb = 0.0f, pu = 0.0f;
        for (int r = lane; r < TF; r += 64) {
            const int e = c * TF + r;
            const float vh = (v[e] - mean) * inv;
            const float m = 1.0f / (1.0f + expf(-(vh * wc + bc)));
            const float uu = u[e], go = dout[e];
            const float dvn = go * (t_act(uu, a.act) - z[e]) * m * (1.0f - m);
            const float duu = go * m * t_dact(uu, a.act);
            dz[e] = go * (1.0f - m);
            du[e] = duu;
            pw += dvn * vh; pb += dvn; pu += duu;
        }
        pw = t_wave_sum(pw); pb = t_wave_sum(pb); pu = t_wave_sum(pu);
        if (lane == 0) {
            a.dnw_part[(long)s * a.Co + c] = pw; a.dnb_part[(long)s * a.Co + c] = pb;
            a.dbias_part[(long)s * 2 * a.Co + c] = pu;
            S1 += (double)(wc * pb); S2 += (double)(wc * pw);
        }
    }
    t_block_sum2(S1, S2, red);
    const float sd = 1.0f / inv - kTEps;
    const float m1 = (float)(S1 / n), m2 = (float)(S2 / n) / (a.eps_mode ? fmaxf(sd, 1e-30f) : sd);
    for (int c = wave; c < a.Co; c += kTW) {
        const float wc = a.nw[c], bc = a.nb[c];
        float pp = 0.0f;
        for (int r = lane; r < TF; r += 64) {
            const int e = c * TF + r;
            const float vh = (v[e] - mean) * inv;
            const float m = 1.0f / (1.0f + expf(-(vh * wc + bc)));
            const float g = dout[e] * (t_act(u[e], a.act) - z[e]) * m * (1.0f - m) * wc;
            const float d = inv * (g - m1) - vh * m2;
            dv[e] = d;
            pp += d;
        }
        pp = t_wave_sum(pp);
        if (lane == 0) a.dbias_part[(long)s * 2 * a.Co + a.Co + c] = pp;
    }
}

// ---- features (CRN.py:463-467) ---------------------------------------------------------------------------------------------------
// spec [S][M][T][F] complex -> feat [S][2M-1][T][F]: |X_m| for every microphone, then ang_0 - ang_m
__global__ __launch_bounds__(256) void k_tfeat(const cf2 *spec, float *feat, int M, int TF, int atan2_phase) {
    const int s = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= TF) return;
    const cf2 *sp = spec + (long)s * M * TF + i;
    float *o = feat + (long)s * (2 * M - 1) * TF + i;
    float ang0 = 0.0f;
    for (int m = 0; m < M; m++) {
        const cf2 v = sp[(long)m * TF];
        o[(long)m * TF] = sqrtf(v.x * v.x + v.y * v.y + 1e-10f);
        const float ang = atan2_phase ? atan2f(v.y, v.x) : atanf(v.y / (v.x + kTEps) + kTEps);
        if (m == 0) ang0 = ang;
        else o[(long)(M + m - 1) * TF] = ang0 - ang;
    }
}

// ---- mask: decompress_cIRM (utility.py:439-442) + complex multiply with the mic-0 spectrum (CRN.py:491-495) ---------------------
// x [S][2][T][F] (last decoder norm output), spec [S][M][T][F] -> Y [S][T][F] complex
__global__ __launch_bounds__(256) void k_tmask_fwd(const float *x, const cf2 *spec, cf2 *Y, int M, int TF) {
    const int s = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= TF) return;
    const float *xp = x + (long)s * 2 * TF + i;
    const float cr = fminf(fmaxf(xp[0], -9.9f), 9.9f), ci = fminf(fmaxf(xp[TF], -9.9f), 9.9f);
    const float mr = -10.0f * logf((10.0f - cr) / (10.0f + cr)), mi = -10.0f * logf((10.0f - ci) / (10.0f + ci));
    const cf2 X = spec[(long)s * M * TF + i];
    Y[(long)s * TF + i] = cf2{mr * X.x - mi * X.y, mi * X.x + mr * X.y};
}
// dY = the raw STFT of (dy / env) (the adjoint of torch.istft up to the irfft weights c_f / N, c_f = 1 at DC / Nyquist, else 2,
// applied here) -> dx [S][2][T][F]; the clamp passes its gradient inside [-9.9, 9.9] only
__global__ __launch_bounds__(256) void k_tmask_bwd(const cf2 *dY, const float *x, const cf2 *spec, float *dx, int M, int T, int F, float inv_nfft) {
    const int TF = T * F;
    const int s = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= TF) return;
    const int f = i % F;
    const float sc = (f == 0 || f == F - 1) ? inv_nfft : 2.0f * inv_nfft;
    cf2 g = dY[(long)s * TF + i];
    g.x *= sc;
    g.y = (f == 0 || f == F - 1) ? 0.0f : g.y * sc;  // C2R ignores Im of DC / Nyquist: no gradient
    const cf2 X = spec[(long)s * M * TF + i];
    const float dmr = g.x * X.x + g.y * X.y, dmi = -g.x * X.y + g.y * X.x;
    const float *xp = x + (long)s * 2 * TF + i;
    float *o = dx + (long)s * 2 * TF + i;
    const float xr = xp[0], xi = xp[TF];
    o[0] = (xr >= -9.9f && xr <= 9.9f) ? dmr * 200.0f / (100.0f - xr * xr) : 0.0f;
    o[TF] = (xi >= -9.9f && xi <= 9.9f) ? dmi * 200.0f / (100.0f - xi * xi) : 0.0f;
}

// ---- utility.over_add on segment-major segment outputs yseg [N][B][K] (utility.py:373-403) + the K/2 strip (CRN.py:587-588) ----
__global__ __launch_bounds__(256) void k_tola_fwd(const float *yseg, float *out, int B, int K, long L, long skip) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= L) return;
    const long P = K / 2, i2 = i + skip, i1 = i2 + P;
    const float v1 = yseg[((2 * (i1 / K)) * (long)B + b) * K + i1 % K];
    const float v2 = yseg[((2 * (i2 / K) + 1) * (long)B + b) * K + i2 % K];
    out[(long)b * L + i] = (v1 + v2) / 2;
}
// adjoint, with the division by the iSTFT overlap-add envelope folded in: g[n][b][k] = 0.5 * dout[b][i(n, k)] / env[k]
__global__ __launch_bounds__(256) void k_tola_bwd(const float *dout, const float *env, float *g, int B, int K, long L, long skip) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y, n = blockIdx.z;
    if (k >= K) return;
    const long P = K / 2;
    const long pos = (long)(n >> 1) * K + k;          // index into the concatenation of this parity's segments
    const long i = (n & 1) ? pos - skip : pos - skip - P;
    const float v = (i >= 0 && i < L) ? 0.5f * dout[(long)b * L + i] / env[k] : 0.0f;
    g[((long)n * B + b) * K + k] = v;
}

// ---- CRN_ELU deltas (CRN_ELU.py:194-252, 335-340, 375-376) -----------------------------------------------------------------------
// Gated 1x1 pair + norm: tg [S][2C][T][F] holds conv_trans(a) in channels [0, C) and conv_gated(a) in [C, 2C);
// p = t * sigmoid(g); y = gLN(p) (CRN_ELU.py:240-241).  Backward: dp by the gLN formula, dt = dp * sigmoid(g), dg = dp * t * s (1 - s);
// per-channel sums of dt / dg = the bias gradients of the pair.
struct TGateArgs {
    const float *tg, *dy;
    float *y, *dtg;
    const float *w, *b;
    float *stats;
    float *dw_part, *db_part, *dbias_part;  // [S][C], [S][C], [S][2C]
    int C, T, F, eps_mode;
    long yS, yC, yT;  // element (s, c, t, f) of y (forward) / dy (backward) at s * yS + c * yC + t * yT + f
};

__global__ __launch_bounds__(kTT) void k_tgate_fwd(TGateArgs a) {
    __shared__ double red[2 * kTW];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int TF = a.T * a.F, n = a.C * TF;
    const float *t = a.tg + (long)s * 2 * n, *g = t + n;
    double sum = 0, dummy = 0;
    for (int e = tid; e < n; e += kTT) sum += (double)(t[e] / (1.0f + expf(-g[e])));
    t_block_sum2(sum, dummy, red);
    const float mean = (float)(sum / n);
    double sq = 0;
    dummy = 0;
    for (int e = tid; e < n; e += kTT) { const float d = t[e] / (1.0f + expf(-g[e])) - mean; sq += (double)(d * d); }
    t_block_sum2(sq, dummy, red);
    const float var = (float)(sq / n);
    const float inv = 1.0f / ((a.eps_mode ? sqrtf(var) : sqrtf(var + kTEps)) + kTEps);
    if (tid == 0) { a.stats[2 * s] = mean; a.stats[2 * s + 1] = inv; }
    float *y = a.y + (long)s * a.yS;
    for (int e = tid; e < n; e += kTT) {
        const int c = e / TF, r = e - c * TF, tt = r / a.F, f = r - tt * a.F;
        y[c * a.yC + tt * a.yT + f] = (t[e] / (1.0f + expf(-g[e])) - mean) * inv * a.w[c] + a.b[c];
    }
}

__global__ __launch_bounds__(kTT) void k_tgate_bwd(TGateArgs a) {
    __shared__ double red[2 * kTW];
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int TF = a.T * a.F, n = a.C * TF;
    const float *t = a.tg + (long)s * 2 * n, *g = t + n;
    const float *dy = a.dy + (long)s * a.yS;
    float *dt = a.dtg + (long)s * 2 * n, *dg = dt + n;
    const float mean = a.stats[2 * s], inv = a.stats[2 * s + 1];
    double S1 = 0, S2 = 0;
    for (int c = wave; c < a.C; c += kTW) {
        const float wc = a.w[c];
        float pw = 0.0f, pb = 0.0f;
        for (int r = lane; r < TF; r += 64) {
            const int e = c * TF + r, tt = r / a.F, f = r - tt * a.F;
            const float ph = (t[e] / (1.0f + expf(-g[e])) - mean) * inv;
            const float d = dy[c * a.yC + tt * a.yT + f];
            pw += d * ph; pb += d;
        }
        pw = t_wave_sum(pw); pb = t_wave_sum(pb);
        if (lane == 0) {
            a.dw_part[(long)s * a.C + c] = pw; a.db_part[(long)s * a.C + c] = pb;
            S1 += (double)(wc * pb); S2 += (double)(wc * pw);
        }
    }
    t_block_sum2(S1, S2, red);
    const float sd = 1.0f / inv - kTEps;
    const float m1 = (float)(S1 / n), m2 = (float)(S2 / n) / (a.eps_mode ? fmaxf(sd, 1e-30f) : sd);
    for (int c = wave; c < a.C; c += kTW) {
        const float wc = a.w[c];
        float pt = 0.0f, pg = 0.0f;
        for (int r = lane; r < TF; r += 64) {
            const int e = c * TF + r, tt = r / a.F, f = r - tt * a.F;
            const float sg = 1.0f / (1.0f + expf(-g[e])), tv = t[e];
            const float ph = (tv * sg - mean) * inv;
            const float dp = inv * (dy[c * a.yC + tt * a.yT + f] * wc - m1) - ph * m2;
            const float vt = dp * sg, vg = dp * tv * sg * (1.0f - sg);
            dt[e] = vt; dg[e] = vg;
            pt += vt; pg += vg;
        }
        pt = t_wave_sum(pt); pg = t_wave_sum(pg);
        if (lane == 0) { a.dbias_part[(long)s * 2 * a.C + c] = pt; a.dbias_part[(long)s * 2 * a.C + a.C + c] = pg; }
    }
}

// da -> dy through a = ELU(y), from the saved activation only: ELU'(y) = 1 (a > 0) or e^y = a + 1; also the per-channel sums of dy
// (the producing convolution's bias gradient) as [S][C] slabs
__global__ __launch_bounds__(kTT) void k_telu_bwd(float *da, const float *aact, float *dpre_part, int C, int TF) {
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = wave; c < C; c += kTW) {
        float pp = 0.0f;
        for (int r = lane; r < TF; r += 64) {
            const long e = ((long)s * C + c) * TF + r;
            const float av = aact[e];
            const float v = da[e] * (av > 0.0f ? 1.0f : av + 1.0f);
            da[e] = v;
            pp += v;
        }
        pp = t_wave_sum(pp);
        if (lane == 0) dpre_part[(long)s * C + c] = pp;
    }
}

// The three 5-channel 5x5 pre-conv blocks (CRN_ELU.py:335-340): Conv2d(C -> C, (5,5), dilation (fd, 1), padding (2 fd, 0)) over
// cat(buffer[4 frames], x): out[co][t][f] = b + sum w[co][ci][kf][kt] * in[ci][t + kt - 4][f + (kf - 2) fd]; rows t < 0 from xprev.
// C <= 8: vector-ALU kernels (0.6 kMAC per position), one thread per output position.  act = 2 stores ELU(out).
struct TPre5Args {
    const float *x, *xprev, *w, *bias, *dy;
    float *y, *dx, *dw_part;
    int C, T, F, fd, act;
};
__global__ __launch_bounds__(256) void k_pre5_fwd(TPre5Args a) {
    __shared__ float ws[8 * 8 * 25 + 8];
    const int s = blockIdx.y, C = a.C, TF = a.T * a.F;
    for (int i = threadIdx.x; i < C * C * 25; i += 256) ws[i] = a.w[i];
    if (threadIdx.x < C) ws[8 * 8 * 25 + threadIdx.x] = a.bias[threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= TF) return;
    const int t = i / a.F, f = i - t * a.F;
    float acc[8];
#pragma unroll
    for (int co = 0; co < 8; co++) acc[co] = co < C ? ws[8 * 8 * 25 + co] : 0.0f;
    for (int ci = 0; ci < C; ci++)
        for (int kt = 0; kt < 5; kt++) {
            int ts = t + kt - 4;
            const float *src = a.x;
            if (ts < 0) { ts += a.T; src = a.xprev; }
            if (!src) continue;
            const float *row = src + (((long)s * C + ci) * a.T + ts) * a.F;
            for (int kf = 0; kf < 5; kf++) {
                const int ff = f + (kf - 2) * a.fd;
                if (ff < 0 || ff >= a.F) continue;
                const float v = row[ff];
#pragma unroll
                for (int co = 0; co < 8; co++)
                    if (co < C) acc[co] += ws[((co * C + ci) * 5 + kf) * 5 + kt] * v;
            }
        }
#pragma unroll
    for (int co = 0; co < 8; co++)
        if (co < C) a.y[((long)s * C + co) * TF + i] = t_act(acc[co], a.act);
}
// dx[ci][t'][f'] = sum w[co][ci][kf][kt] * dy[co][t' - kt + 4][f' - (kf - 2) fd]   (rows beyond the segment contribute nothing: the
// history is detached, CRN_ELU.py:243)
__global__ __launch_bounds__(256) void k_pre5_dx(TPre5Args a) {
    __shared__ float ws[8 * 8 * 25];
    const int s = blockIdx.y, C = a.C, TF = a.T * a.F;
    for (int i = threadIdx.x; i < C * C * 25; i += 256) ws[i] = a.w[i];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= TF) return;
    const int t = i / a.F, f = i - t * a.F;
    float acc[8];
#pragma unroll
    for (int ci = 0; ci < 8; ci++) acc[ci] = 0.0f;
    for (int co = 0; co < C; co++)
        for (int kt = 0; kt < 5; kt++) {
            const int to = t - kt + 4;
            if (to >= a.T) continue;
            const float *row = a.dy + (((long)s * C + co) * a.T + to) * a.F;
            for (int kf = 0; kf < 5; kf++) {
                const int ff = f - (kf - 2) * a.fd;
                if (ff < 0 || ff >= a.F) continue;
                const float v = row[ff];
#pragma unroll
                for (int ci = 0; ci < 8; ci++)
                    if (ci < C) acc[ci] += ws[((co * C + ci) * 5 + kf) * 5 + kt] * v;
            }
        }
#pragma unroll
    for (int ci = 0; ci < 8; ci++)
        if (ci < C) a.dx[((long)s * C + ci) * TF + i] = acc[ci];
}
// dW[co][ci][kf][kt] partial of ONE stream: block (pair = co * C + ci, s); every thread keeps the 25 tap sums of its positions
__global__ __launch_bounds__(256) void k_pre5_dw(TPre5Args a) {
    __shared__ float red[4][25];
    const int s = blockIdx.y, C = a.C, TF = a.T * a.F;
    const int co = blockIdx.x / C, ci = blockIdx.x - co * C;
    float acc[25];
#pragma unroll
    for (int k = 0; k < 25; k++) acc[k] = 0.0f;
    const float *dyc = a.dy + ((long)s * C + co) * TF;
    for (int i = threadIdx.x; i < TF; i += 256) {
        const int t = i / a.F, f = i - t * a.F;
        const float g = dyc[i];
#pragma unroll
        for (int kt = 0; kt < 5; kt++) {
            int ts = t + kt - 4;
            const float *src = a.x;
            if (ts < 0) { ts += a.T; src = a.xprev; }
            if (!src) continue;
            const float *row = src + (((long)s * C + ci) * a.T + ts) * a.F;
#pragma unroll
            for (int kf = 0; kf < 5; kf++) {
                const int ff = f + (kf - 2) * a.fd;
                if (ff >= 0 && ff < a.F) acc[kf * 5 + kt] += g * row[ff];
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 25; k++) {
        const float v = t_wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 25) a.dw_part[((long)s * C * C + blockIdx.x) * 25 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void k_tadd3(float *dst, const float *a, const float *b, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = a[i] + b[i];
}

__global__ __launch_bounds__(256) void k_tadd(float *dst, const float *src, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

// h_{s-1} rows for the recurrent weight gradient: hp[row(b, s)] = s == 0 ? h0[b] : out[row(b, s - 1)]
__global__ __launch_bounds__(256) void k_gru_hprev(const float *out, const float *h0, float *hp, int B, int T, int H, int Tseg, long ldN, long ldB) {
    const int j = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y, b = blockIdx.z;
    if (j >= H) return;
    const int n = s / Tseg;
    const long row = (long)n * ldN + (long)b * ldB + (s - n * Tseg);
    float v;
    if (s == 0) v = h0[(long)b * H + j];
    else { const int n1 = (s - 1) / Tseg; v = out[((long)n1 * ldN + (long)b * ldB + (s - 1 - n1 * Tseg)) * H + j]; }
    hp[row * H + j] = v;
}

// weight re-arrangement for k_conv_igemm: w element (co, ci, kf, kt) at co * sCo + ci * sCi + kf * 3 + kt (nkk = 15) or co * sCo + ci * sCi
// (1x1) -> [nchunk][ntap][CC][CoPad] fp32, zero padded
struct TArrangeArgs {
    const float *w;
    float *out;
    long sCo, sCi;
    int Co, Ci, ntap, CC, nchunk, CoPad, one_by_one;
    int tap_kf[15], tap_kt[15];
};
__global__ __launch_bounds__(256) void k_arrange_w(TArrangeArgs a) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)a.nchunk * a.ntap * a.CC * a.CoPad;
    if (i >= total) return;
    const int co = (int)(i % a.CoPad);
    long r = i / a.CoPad;
    const int cc = (int)(r % a.CC); r /= a.CC;
    const int tap = (int)(r % a.ntap);
    const int chunk = (int)(r / a.ntap);
    const int ci = chunk * a.CC + cc;
    float v = 0.0f;
    if (co < a.Co && ci < a.Ci) v = a.w[co * a.sCo + ci * a.sCi + (a.one_by_one ? 0 : a.tap_kf[tap] * 3 + a.tap_kt[tap])];
    a.out[i] = v;
}

}  // namespace se
