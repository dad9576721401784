// norm.hip.h - per-stream global layer norm and the pointwise stages fused around it.
//
// Reference: GlobalLayerNorm(time=False).forward (CRN.py:135-149): per sample, mean and BIASED variance
// over every non-batch element (two passes, like the reference), y = (x - mu) / (sqrt(var + 1e-8) + 1e-8) * w + b.
// One workgroup (1024 threads = 16 wavefronts) owns one stream's tensor (<= ~180 KB, L2 resident after
// the first pass); reductions are wavefront shuffles + one LDS hop, combined in double.
// Fused variants:
//   k_featurize  - |X|, arctan phase differences (CRN.py:463-467)
//   k_gln           - exact two-pass norm + affine + re-layout, one workgroup per stream (used for the FC output, whose
//                     producer is a GEMM without per-stream partial statistics)
//   k_gln_ew        - norm + affine (+ residual add, + re-layout) from the producing conv's partial statistics
//   k_dec_blend_ew  - decoder skip gate: m = sigmoid(gLN(conv_mask(res))), out = m*act(conv_res(res)) + (1-m)*pad(gLN(y))
//                     (CRN.py:387-396)
//   k_final_mask_ew - gLN of the last decoder block, decompress_cIRM (utility.py:439-442), complex multiply
//                     with the mic-0 spectrum (CRN.py:491-495)
#pragma once
#include <hip/hip_runtime.h>
#include "fft_lds.h"

namespace se {

constexpr float kEps = 1e-8f;  // CRN.py:11

// gLN denominator: sqrt(var + eps) + eps (CRN.py:149, CRN_ELU.py:51) or sqrt(var) + eps (distillation_crn.py:51)
__device__ __forceinline__ float gln_inv(float var, int eps_outside_only) {
    return 1.0f / ((eps_outside_only ? sqrtf(var) : sqrtf(var + kEps)) + kEps);
}

__device__ inline double block_sum(double v, double *red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();  // protect red from a previous use
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0;
    for (int i = 0; i < nw; i++) s += red[i];
    return s;
}

// mean and 1/(sqrt(var+eps)+eps) of n contiguous floats (two-pass, biased variance)
__device__ inline void stream_stats(const float *x, long n, double *red, float &mean, float &inv, int eps_mode = 0) {
    float part = 0.0f;
    for (long i = threadIdx.x; i < n; i += blockDim.x) part += x[i];
    const double s = block_sum((double)part, red);
    mean = (float)(s / (double)n);
    float p2 = 0.0f;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
        const float d = x[i] - mean;
        p2 += d * d;
    }
    const double q = block_sum((double)p2, red);
    const float var = (float)(q / (double)n);
    inv = gln_inv(var, eps_mode);
}

// ---- featurise: spec (b, m, t, f) -> feat [B][2M-1][T][F] --------------------------------------
struct FeatArgs {
    const cf2 *spec;
    long sB, sM, sT, sF;  // strides in cf2 units
    float *feat;
    int M, T, F;
    int atan2_phase;  // 0: arctan(im/(re+eps)+eps) (CRN.py:464, distillation_crn.py:340); 1: atan2(im, re) (CRN_ELU.py:370)
};

#if !defined(SE_AUX_KERNELS) && !defined(SE_NO_NORM_KERNELS)
__global__ void k_featurize(FeatArgs a) {
    const int b = blockIdx.y;
    const int TF = a.T * a.F;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= TF) return;
    const int t = i / a.F, f = i - t * a.F;
    const cf2 *s = a.spec + (long)b * a.sB + (long)t * a.sT + (long)f * a.sF;
    float *o = a.feat + (long)b * (2 * a.M - 1) * TF + i;
    float ang0 = 0.0f;
    for (int m = 0; m < a.M; m++) {
        const cf2 v = s[(long)m * a.sM];
        o[(long)m * TF] = sqrtf(v.x * v.x + v.y * v.y + 1e-10f);
        const float ang = a.atan2_phase ? atan2f(v.y, v.x) : atanf(v.y / (v.x + kEps) + kEps);
        if (m == 0) ang0 = ang;
        else o[(long)(a.M + m - 1) * TF] = ang0 - ang;
    }
}

#endif  // norm kernels

// ---- gLN with optional re-layout ---------------------------------------------------------------
struct GlnArgs {
    const float *x;  // per stream n contiguous floats
    float *y;
    const float *w, *b;
    long n;
    int mode;  // 0: [C][T][F] -> same, affine per C | 1: [C][T][F] -> [T][C*F], affine per C
               // 2: [T][D=C*F] -> [C][T][F], affine per D (GlobalLayerNorm(last=True), CRN.py:127-129)
    int C, T, F;
    int eps_mode;
};

#if !defined(SE_AUX_KERNELS) && !defined(SE_NO_NORM_KERNELS)
__global__ __launch_bounds__(1024) void k_gln(GlnArgs a) {
    __shared__ double red[16];
    const float *x = a.x + (long)blockIdx.x * a.n;
    float *y = a.y + (long)blockIdx.x * a.n;
    float mean, inv;
    stream_stats(x, a.n, red, mean, inv, a.eps_mode);
    const int TF = a.T * a.F, F = a.F, T = a.T, C = a.C;
    if (a.mode == 0) {
        for (long i = threadIdx.x; i < a.n; i += blockDim.x) {
            const int c = (int)(i / TF);
            y[i] = (x[i] - mean) * inv * a.w[c] + a.b[c];
        }
    } else if (a.mode == 1) {
        for (long i = threadIdx.x; i < a.n; i += blockDim.x) {
            const int c = (int)(i / TF), r = (int)(i - (long)c * TF), t = r / F, f = r - t * F;
            y[((long)t * C + c) * F + f] = (x[i] - mean) * inv * a.w[c] + a.b[c];
        }
    } else {
        const int D = C * F;
        for (long i = threadIdx.x; i < a.n; i += blockDim.x) {
            const int t = (int)(i / D), d = (int)(i - (long)t * D), c = d / F, f = d - c * F;
            y[((long)c * T + t) * F + f] = (x[i] - mean) * inv * a.w[d] + a.b[d];
        }
    }
}

#endif  // norm kernels

// ---- cIRM decompression (utility.py:439-442) --------------------------------------------------------------
__device__ inline float decompress_cirm(float m) {
    m = m >= 9.9f ? 9.9f : (m <= -9.9f ? -9.9f : m);
    return -10.0f * logf((10.0f - m) / (10.0f + m));
}

// ---- elementwise variants: statistics come from the producing convolution's per-workgroup partials ----------
// (one pass over the tensor instead of three, full-chip grid instead of one workgroup per stream)
struct SlabStats { const float *slab; int nslot; long n; int eps_mode; };  // slab [B][nslot][2] = (sum, sum of squares)

__device__ __forceinline__ void slab_mean_inv(const SlabStats &st, int b, float *sm /*[2] shared*/, float &mean, float &inv) {
    if (threadIdx.x < 64) {
        double s = 0, q = 0;
        const float *p = st.slab + (long)b * st.nslot * 2;
        for (int i = threadIdx.x; i < st.nslot; i += 64) { s += (double)p[2 * i]; q += (double)p[2 * i + 1]; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); q += __shfl_down(q, off, 64); }
        if (threadIdx.x == 0) {
            const double m = s / (double)st.n;
            double var = q / (double)st.n - m * m;
            if (var < 0) var = 0;
            sm[0] = (float)m;
            sm[1] = gln_inv((float)var, st.eps_mode);
        }
    }
    __syncthreads();
    mean = sm[0];
    inv = sm[1];
}

struct GlnEwArgs {
    const float *x;
    float *y;
    const float *w, *b;
    SlabStats st;
    int mode;  // 0: [C][T][F] -> same | 1: [C][T][F] -> [T][C*F]
    int C, T, F;
    const float *res;  // mode 0 only: y = gLN(x) + res (the `m(x) + x` of the preconv blocks, CRN_ELU.py:375-376)
};

// VW = 4: float4 path (n % 4 == 0, so every stream's base stays 16-B aligned); VW = 1: scalar path for odd sizes
// (the 5-channel preconv tensors of CRN_ELU: 5*21*201 elements).
#if !defined(SE_AUX_KERNELS) && !defined(SE_NO_NORM_KERNELS)
template <int VW>
__global__ __launch_bounds__(256) void k_gln_ew(GlnEwArgs a) {
    __shared__ float sm[2];
    const int b = blockIdx.y;
    float mean, inv;
    slab_mean_inv(a.st, b, sm, mean, inv);
    const long n = a.st.n;
    const float *x = a.x + (long)b * n;
    float *y = a.y + (long)b * n;
    const float *res = a.res ? a.res + (long)b * n : nullptr;
    const int TF = a.T * a.F, F = a.F, C = a.C;
    const float invTF = 1.0f / (float)TF;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VW; i < n; i += (long)gridDim.x * blockDim.x * VW) {
        float vv[VW], o[VW];
        int cc[VW];
        if (VW == 4) {
            const float4 v = *reinterpret_cast<const float4 *>(x + i);
            vv[0] = v.x; vv[VW > 1 ? 1 : 0] = v.y; vv[VW > 2 ? 2 : 0] = v.z; vv[VW > 3 ? 3 : 0] = v.w;
        } else {
            vv[0] = x[i];
        }
#pragma unroll
        for (int k = 0; k < VW; k++) {
            int c = (int)(((float)(i + k) + 0.5f) * invTF);  // (i+k) / TF, exact for n < 2^22
            c = min(c, C - 1);
            cc[k] = c;
            o[k] = (vv[k] - mean) * inv * a.w[c] + a.b[c];
        }
        if (a.mode == 0) {
            if (VW == 4) {
                if (res) {
                    const float4 r4 = *reinterpret_cast<const float4 *>(res + i);
                    o[0] += r4.x; o[VW > 1 ? 1 : 0] += r4.y; o[VW > 2 ? 2 : 0] += r4.z; o[VW > 3 ? 3 : 0] += r4.w;
                }
                *reinterpret_cast<float4 *>(y + i) = make_float4(o[0], o[VW > 1 ? 1 : 0], o[VW > 2 ? 2 : 0], o[VW > 3 ? 3 : 0]);
            } else {
                y[i] = o[0] + (res ? res[i] : 0.0f);
            }
        } else {
#pragma unroll
            for (int k = 0; k < VW; k++) {
                const int r = (int)(i + k) - cc[k] * TF, t = r / F, f = r - t * F;
                y[((long)t * C + cc[k]) * F + f] = o[k];
            }
        }
    }
}

#endif  // norm kernels

// R layout (conv_p.hip.h: [b][octet][t*F + f][8] fp32) -> normalised [T][C*F] fp32 rows, the A operand of the bottleneck's input
// GEMM (CRN.py:476-478: reshape [B, C, F, T] -> [B, C*F, T] -> permute); one thread per (octet, position)
#if !defined(SE_AUX_KERNELS) && !defined(SE_NO_NORM_KERNELS)
__global__ __launch_bounds__(256) void k_gln_r2t(GlnEwArgs a, long x_stream) {
    __shared__ float sm[2];
    const int b = blockIdx.y;
    float mean, inv;
    slab_mean_inv(a.st, b, sm, mean, inv);
    const int TF = a.T * a.F, C8 = (a.C + 7) / 8;
    const float *x = a.x + (long)b * x_stream;
    float *y = a.y + (long)b * a.st.n;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < C8 * TF; i += gridDim.x * 256) {
        const int o = i / TF, pos = i - o * TF, t = pos / a.F, f = pos - t * a.F;
        const float4 *src = reinterpret_cast<const float4 *>(x + ((long)o * TF + pos) * 8);
        const float4 u0 = src[0], u1 = src[1];
        const float v[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int ch = o * 8 + c;
            if (ch < a.C) y[((long)t * a.C + ch) * a.F + f] = (v[c] - mean) * inv * a.w[ch] + a.b[ch];
        }
    }
}
#endif  // norm kernels

struct BlendEwArgs {
    const float *y, *uv;
    float *out;
    const float *nw, *nb, *mnw, *mnb;
    SlabStats sy, su;
    int Co, T, Fo, Fr;
};

#if !defined(SE_AUX_KERNELS) && !defined(SE_NO_NORM_KERNELS)
__global__ __launch_bounds__(256) void k_dec_blend_ew(BlendEwArgs a) {
    __shared__ float sm[4];
    const int b = blockIdx.y;
    float my, iy, mu, iu;
    slab_mean_inv(a.sy, b, sm, my, iy);
    slab_mean_inv(a.su, b, sm + 2, mu, iu);
    const long ny = (long)a.Co * a.T * a.Fo, nu = (long)a.Co * a.T * a.Fr;
    const float *y = a.y + b * ny;
    const float *u = a.uv + (long)b * 2 * nu;
    const float *v = u + nu;
    float *o = a.out + b * nu;
    const int TFr = a.T * a.Fr;
    const float invTFr = 1.0f / (float)TFr, invFr = 1.0f / (float)a.Fr;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < nu; i += (long)gridDim.x * blockDim.x * 4) {
        const float4 u4 = *reinterpret_cast<const float4 *>(u + i);
        const float4 v4 = *reinterpret_cast<const float4 *>(v + i);
        const float uu[4] = {u4.x, u4.y, u4.z, u4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
        float r4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int c = min((int)(((float)(i + k) + 0.5f) * invTFr), a.Co - 1);
            const int r = (int)(i + k) - c * TFr;
            int t = min((int)(((float)r + 0.5f) * invFr), a.T - 1);
            const int f = r - t * a.Fr;
            const float yv = f < a.Fo ? (y[((long)c * a.T + t) * a.Fo + f] - my) * iy * a.nw[c] + a.nb[c] : 0.0f;
            const float un = (uu[k] - mu) * iu * a.mnw[c] + a.mnb[c];
            const float m = 1.0f / (1.0f + expf(-un));
            r4[k] = m * vv[k] + (1.0f - m) * yv;
        }
        *reinterpret_cast<float4 *>(o + i) = make_float4(r4[0], r4[1], r4[2], r4[3]);
    }
}

#endif  // norm kernels

struct MaskEwArgs {
    const float *y;
    const float *nw, *nb;
    SlabStats st;
    const cf2 *spec;
    long sB, sT, sF;
    cf2 *out;
    long oB, oT, oF;
    int T, F;
};

void launch_k_final_mask_ew(dim3 grid, hipStream_t st, const MaskEwArgs &a);  // defined in se_aux.hip

#ifdef SE_AUX_KERNELS
__global__ __launch_bounds__(256) void k_final_mask_ew(MaskEwArgs a) {
    __shared__ float sm[2];
    const int b = blockIdx.y;
    float mean, inv;
    slab_mean_inv(a.st, b, sm, mean, inv);
    const int TF = a.T * a.F;
    const float *y = a.y + (long)b * 2 * TF;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < TF; i += gridDim.x * blockDim.x) {
        const int t = i / a.F, f = i - t * a.F;
        const float mr = decompress_cirm((y[i] - mean) * inv * a.nw[0] + a.nb[0]);
        const float mi = decompress_cirm((y[TF + i] - mean) * inv * a.nw[1] + a.nb[1]);
        const cf2 n = a.spec[(long)b * a.sB + (long)t * a.sT + (long)f * a.sF];
        a.out[(long)b * a.oB + (long)t * a.oT + (long)f * a.oF] = cf2{mr * n.x - mi * n.y, mi * n.x + mr * n.y};
    }
}

#endif  // SE_AUX_KERNELS

}  // namespace se
