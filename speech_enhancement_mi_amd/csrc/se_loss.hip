// se_loss.hip - device-resident training-loss kernels behind the se_loss_* C ABI (include/se_engine.h).
//
// SI-SNR term of TemporalCRN.compute_loss (reference CRN.py:610, utility.cal_si_snr utility.py:207-223): for every
// utterance i, on its first len[i] samples, with the means removed,
//     s_t = <s, r> r / (|r|^2 + eps),   SI-SNR_i = 20 log10(eps + |s_t| / (|s - s_t| + eps)).
// The reference loops over the batch in Python with ~10 small torch ops per utterance; here ONE workgroup per utterance
// reduces the moments (three passes over data that sits in L2, partial sums in double, fixed reduction order: results are
// bit-reproducible) and the backward is a closed-form elementwise kernel using the saved per-utterance scalars.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/se_engine.h"

namespace {

constexpr double kEps = 1e-8;  // utility.py:207 default eps

__device__ inline double block_sum1024(double v, double *red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0;
    for (int i = 0; i < 16; i++) s += red[i];
    return s;
}

// stats[b*8 + ...] = {mean_s, mean_r, a = dot/(R+eps), R = |r~|^2, T = |s_t|, E = |s~ - s_t|, n, dot}
__global__ __launch_bounds__(1024) void k_sisnr_fwd(const float *sep, const float *src, const int64_t *lens, long L, float *per, double *stats) {
    __shared__ double red[16];
    const int b = blockIdx.x;
    const long n = lens[b] < L ? (lens[b] > 0 ? lens[b] : 0) : L;
    const float *s = sep + (long)b * L, *r = src + (long)b * L;
    double ss = 0, sr = 0;
    for (long k = threadIdx.x; k < n; k += 1024) { ss += (double)s[k]; sr += (double)r[k]; }
    const double ms = n ? block_sum1024(ss, red) / (double)n : 0.0;
    const double mr = n ? block_sum1024(sr, red) / (double)n : 0.0;
    double R = 0, dot = 0;
    for (long k = threadIdx.x; k < n; k += 1024) {
        const double sc = (double)s[k] - ms, rc = (double)r[k] - mr;
        R += rc * rc;
        dot += sc * rc;
    }
    R = block_sum1024(R, red);
    dot = block_sum1024(dot, red);
    const double a = dot / (R + kEps);
    double E2 = 0;
    for (long k = threadIdx.x; k < n; k += 1024) {
        const double e = ((double)s[k] - ms) - a * ((double)r[k] - mr);
        E2 += e * e;
    }
    E2 = block_sum1024(E2, red);
    if (threadIdx.x == 0) {
        const double T = fabs(a) * sqrt(R), E = sqrt(E2);
        per[b] = (float)(20.0 * log10(kEps + T / (E + kEps)));
        double *o = stats + (long)b * 8;
        o[0] = ms; o[1] = mr; o[2] = a; o[3] = R; o[4] = T; o[5] = E; o[6] = (double)n; o[7] = dot;
    }
}

// grad[b][k] = gscale * d SI-SNR_b / d s[k]  (zero beyond len[b]);  see the derivation in losses.py / DESIGN.md
__global__ __launch_bounds__(256) void k_sisnr_bwd(const float *sep, const float *src, const int64_t *lens, long L, const double *stats,
                                                   const float *gscale, float *grad) {
    const int b = blockIdx.y;
    const double *st = stats + (long)b * 8;
    const double ms = st[0], mr = st[1], a = st[2], R = st[3], T = st[4], E = st[5], dot = st[7];
    const long n = (long)st[6];
    const double q = T / (E + kEps);
    const double dfdq = (20.0 / 2.302585092994046) / (kEps + q) * (double)gscale[0];
    // dT/ds~_k = c1 * r~_k ;  dE/ds~_k = (e_k - c3 * r~_k) / E  (0 when E == 0, the subgradient torch.norm uses)
    const double sgn = dot > 0 ? 1.0 : (dot < 0 ? -1.0 : 0.0);
    const double c1 = sgn * sqrt(R) / (R + kEps);
    const double c3 = (dot - a * R) / (R + kEps);
    const double wT = dfdq / (E + kEps), wE = E > 0 ? dfdq * T / ((E + kEps) * (E + kEps)) / E : 0.0;
    const float *s = sep + (long)b * L, *r = src + (long)b * L;
    float *g = grad + (long)b * L;
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < L; k += (long)gridDim.x * 256) {
        if (k >= n) { g[k] = 0.0f; continue; }
        const double rc = (double)r[k] - mr, e = ((double)s[k] - ms) - a * rc;
        g[k] = (float)(wT * c1 * rc - wE * (e - c3 * rc));
    }
}

}  // namespace

extern "C" {

int se_loss_sisnr_fwd(const float *separated, const float *source, const int64_t *lens, int batch, int64_t length, float *per_utt,
                      double *stats, void *stream) {
    if (!separated || !source || !lens || !per_utt || !stats || batch <= 0 || length <= 0) return SE_ERR_ARG;
    hipLaunchKernelGGL(k_sisnr_fwd, dim3(batch), dim3(1024), 0, static_cast<hipStream_t>(stream), separated, source, lens, (long)length, per_utt, stats);
    return hipGetLastError() == hipSuccess ? SE_OK : SE_ERR_HIP;
}

int se_loss_sisnr_bwd(const float *separated, const float *source, const int64_t *lens, int batch, int64_t length, const double *stats,
                      const float *gscale, float *grad, void *stream) {
    if (!separated || !source || !lens || !stats || !gscale || !grad || batch <= 0 || length <= 0) return SE_ERR_ARG;
    const unsigned gx = (unsigned)((length + 1023) / 1024);
    hipLaunchKernelGGL(k_sisnr_bwd, dim3(gx, batch), dim3(256), 0, static_cast<hipStream_t>(stream), separated, source, lens, (long)length, stats, gscale, grad);
    return hipGetLastError() == hipSuccess ? SE_OK : SE_ERR_HIP;
}

}  // extern "C"
