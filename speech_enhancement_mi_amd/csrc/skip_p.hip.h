// skip_p.hip.h - the decoder's skip gate as ONE streaming kernel per level (reference CRN.py:387-396):
//
//     m   = sigmoid(gLN(residualmask(res)))          1x1 conv of the encoder output `res`, per-stream global layer norm
//     out = m * act(residual(res)) + (1 - m) * pad(gLN(act(deconv)))
//
// The global layer norm of the residualmask output needs that convolution's per-stream statistics BEFORE any output can be
// gated.  The first form of the plane path ran a statistics-only k_conv_p launch followed by a second k_conv_p launch with
// the gate in its epilogue: two LDS-staged launches of a few MFMAs each, 7.5 rounds of workgroups whose DMA latency,
// barriers and fragment fetches were all exposed (100 us per level for 13 us of memory traffic).  A 1x1 convolution has no
// halo and (with <= 32 output rows per M tile) nothing to share through LDS, so here ONE workgroup owns one stream and
// streams it twice through registers:
//     pass 1   B fragments straight from the P layout (16-byte pieces, consecutive lanes = consecutive positions),
//              residualmask rows only -> (sum, sum of squares) -> workgroup reduction -> mean / inverse deviation
//     pass 2   the same fragments again (L2 / MALL hits), residualmask + residual rows -> gate -> P layout of the block output
// Weights (<= 48 KB) sit in LDS in fragment order; no barrier inside either pass.
#pragma once
#include <hip/hip_runtime.h>

#include "conv_p.hip.h"

namespace se {


// NW = waves per workgroup: 16 (1024 threads, 128 VGPRs) for KP <= 2, 8 for KP = 4 - the kernel is a latency-bound stream
// (one workgroup per CU), so bytes in flight = waves x fragments in flight decide its rate
template <int PL, int KP, int NW>
__global__ __launch_bounds__(NW * 64) void k_skip_p(SkipPArgs a) {
    extern __shared__ __align__(16) uint4 wl[];  // [2*MTh][KP][PL][64] weight fragments, then cst[6][Cp] floats
    __shared__ float sm[4];
    __shared__ double red[2][NW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.x;
    const int nfrag = 2 * a.MTh * KP * PL * 64;
    for (int i = tid; i < nfrag; i += NW * 64) wl[i] = a.wx[i];
    const int Cp = a.MTh * 32;
    float *cst = reinterpret_cast<float *>(wl + nfrag);  // per-channel constants in LDS: global loads between the gate's stores
    for (int i = tid; i < 6 * Cp; i += NW * 64) cst[i] = a.cst[i];  // could not be hoisted (they may alias), LDS reads can
    float my, iy;
    slab_mean_inv(a.sy, b, sm, my, iy);  // (contains a barrier: the weights are in LDS afterwards)
    const uint4 *xb = a.x + (long)b * a.x_stream;
    const int tiles = (a.TF + 31) >> 5;
    const long plane = a.TF;  // uint4 per plane of one octet

    // B fragments of one position tile: KP K steps x PL planes; lane half h of step kp holds octet 2 kp + h
    auto load_b = [&](int tile, uint4 (&bf)[KP][PL]) {
        const int p = min(tile * 32 + l31, a.TF - 1);
#pragma unroll
        for (int kp = 0; kp < KP; kp++) {
            const int o = min(2 * kp + half, a.C8 - 1);  // a missing odd octet reads a valid one (its weights are zero)
#pragma unroll
            for (int pl = 0; pl < PL; pl++) bf[kp][pl] = xb[((long)o * PL + pl) * plane + p];
        }
    };
    auto mma_tile = [&](int mt, const uint4 (&bf)[KP][PL]) {
        f32x16 c;
#pragma unroll
        for (int r = 0; r < 16; r++) c[r] = 0.0f;
#pragma unroll
        for (int kp = 0; kp < KP; kp++) {
            const uint4 *wf = wl + ((long)(mt * KP + kp) * PL) * 64 + l31 * 2 + half;
            if (PL >= 2) {
                const bf16x8 a0 = __builtin_bit_cast(bf16x8, wf[0]), a1 = __builtin_bit_cast(bf16x8, wf[PL > 1 ? 64 : 0]);
                const bf16x8 b0 = __builtin_bit_cast(bf16x8, bf[kp][0]), b1 = __builtin_bit_cast(bf16x8, bf[kp][PL > 1 ? 1 : 0]);
                if (PL == 3) {
                    const bf16x8 a2 = __builtin_bit_cast(bf16x8, wf[PL > 2 ? 128 : 0]), b2 = __builtin_bit_cast(bf16x8, bf[kp][PL > 2 ? 2 : 0]);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c, 0, 0, 0);
            } else {
                typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, wf[0]), __builtin_bit_cast(h8, bf[kp][0]), c, 0, 0, 0);
            }
        }
        return c;
    };
    // wl fragment layout per (mt, kp, plane): 64 uint4 = [row 32][k half 2]; lane (row l31, half) reads l31*2 + half

    // ---- pass 1: statistics of residualmask(res) ----
    double s1 = 0, s2 = 0;
    {
        uint4 bcur[KP][PL], bnxt[KP][PL];
        if (wave < tiles) load_b(wave, bcur);
        for (int tile = wave; tile < tiles; tile += NW) {
            if (tile + NW < tiles) load_b(tile + NW, bnxt);
            const bool pos_ok = tile * 32 + l31 < a.TF;
            float ts = 0.0f, tq = 0.0f;
            for (int mt = 0; mt < a.MTh; mt++) {
                const f32x16 c = mma_tile(mt, bcur);
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int ch = mt * 32 + 16 * half + r;
                    const float v = c[r] + cst[ch];
                    if (pos_ok && ch < a.C) { ts += v; tq += v * v; }
                }
            }
            s1 += (double)ts;
            s2 += (double)tq;
#pragma unroll
            for (int kp = 0; kp < KP; kp++)
#pragma unroll
                for (int pl = 0; pl < PL; pl++) bcur[kp][pl] = bnxt[kp][pl];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); }
    if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
    __syncthreads();
    if (tid == 0) {
        double ts = 0, tq = 0;
        for (int i = 0; i < NW; i++) { ts += red[0][i]; tq += red[1][i]; }
        const double n = (double)a.C * a.TF, m = ts / n;
        double var = tq / n - m * m;
        if (var < 0) var = 0;
        sm[2] = (float)m;
        sm[3] = gln_inv((float)var, a.eps_mode);
    }
    __syncthreads();
    const float mu = sm[2], iu = sm[3];

    // ---- pass 2: gate ----
    const float invF = 1.0f / (float)a.Fr;
    const float *ydb = a.ydec + (long)b * a.y_stream;
    uint4 *ob = a.out + (long)b * a.out_stream;
    uint4 bcur[KP][PL], bnxt[KP][PL];
    if (wave < tiles) load_b(wave, bcur);
    for (int tile = wave; tile < tiles; tile += NW) {
        if (tile + NW < tiles) load_b(tile + NW, bnxt);
        const int p = tile * 32 + l31;
        const bool pos_ok = p < a.TF;
        const int pc = min(p, a.TF - 1);
        const int t = (int)(((float)pc + 0.5f) * invF), f = pc - t * a.Fr;
        const bool has_y = f < a.Fo;
        const long ypos = (long)t * 2 * a.Fh + (f & 1) * a.Fh + (f >> 1);
        for (int mt = 0; mt < a.MTh; mt++) {
            const f32x16 cm = mma_tile(mt, bcur), cr = mma_tile(a.MTh + mt, bcur);
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const int oct = mt * 4 + 2 * half + hh;  // registers 8 hh .. 8 hh + 7 of this lane
                if (oct * 8 >= a.C) continue;
                float yv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                if (has_y) {
                    const float4 *src = reinterpret_cast<const float4 *>(ydb + ((long)oct * a.T * 2 * a.Fh + ypos) * 8);
                    const float4 u0 = src[0], u1 = src[1];
                    yv[0] = u0.x; yv[1] = u0.y; yv[2] = u0.z; yv[3] = u0.w; yv[4] = u1.x; yv[5] = u1.y; yv[6] = u1.z; yv[7] = u1.w;
                }
                float o[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int r = hh * 8 + q, ch = oct * 8 + q;
                    const float u = cm[r] + cst[ch];
                    const float v = convp_act(cr[r] + cst[Cp + ch], a.act);
                    const float yn = has_y ? (yv[q] - my) * iy * cst[2 * Cp + ch] + cst[3 * Cp + ch] : 0.0f;
                    const float un = (u - mu) * iu * cst[4 * Cp + ch] + cst[5 * Cp + ch];
                    const float g = 1.0f / (1.0f + expf(-un));
                    o[q] = ch < a.C ? g * v + (1.0f - g) * yn : 0.0f;
                }
                if (pos_ok) split_store8<PL>(o, ob + (long)oct * PL * plane + p, plane);
            }
        }
#pragma unroll
        for (int kp = 0; kp < KP; kp++)
#pragma unroll
            for (int pl = 0; pl < PL; pl++) bcur[kp][pl] = bnxt[kp][pl];
    }
}

}  // namespace se
