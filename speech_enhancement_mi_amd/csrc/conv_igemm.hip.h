// conv_igemm.hip.h - implicit-GEMM (dilated, strided / transposed / 1x1) 2-D convolution on the fp32
// matrix cores of gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, SURVEY.md F9 - the path is FLOP-bound).
//
// One kernel covers every convolution of the CRN (reference CRN.py:290-401):
//   * TemporalConv2d       Conv2d(k=(5,3), stride (2,1), pad (2,0), dil (1,d)) over cat(buffer, x)  (CRN.py:321-338)
//   * TemporalConvTranspose2d ConvTranspose2d(same geometry) keeping the LAST T columns (CRN.py:379-386),
//                          split by output-frequency parity into two stride-1 tap sets (3x3 and 2x3 taps)
//   * the decoder's two 1x1 convs residualmask / residual (CRN.py:372,375,395-396), stacked into one GEMM.
//
// GEMM view: Out[co, p] = sum_{tap, ci} W[tap][ci][co] * X[ci][row(p)+rowoff(tap)][s*m(p)+coloff(tap)]
//   M = Cout (A operand = weights), N = flattened output positions p = (t, m) of ONE stream,
//   K = taps x Cin, walked in Cin-chunks of CC channels.
// Activations live in HBM as [B][C][T][F] (F innermost so patch rows are contiguous 800-B runs).
// Per chunk the workgroup stages (a) the input patch [CC][rows][cols] with zero halo - the causal history
// rows t<0 come from the PREVIOUS step's copy of the same tensor (ping-pong buffers replace the
// reference's explicit `buffer`, CRN.py:325-337) - and (b) the weight slab [taps][CC][CoutPad] into LDS;
// im2col happens at LDS-read time (one ds_read_b32 per B fragment at lane_base + tap offset).
// Each of the 4 waves owns one 32-row Cout tile and up to NT 32-position column tiles (register
// blocking: one A fragment feeds NT MFMAs).
#pragma once
#include <hip/hip_runtime.h>

namespace se {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxTaps = 15;

struct ConvArgs {
    const float *x;      // [B][Ci][T][Fi] current step
    const float *xprev;  // same tensor of the previous step (rows t<0), or nullptr
    const float *w;      // [nchunk][ntap][CC][CoPad] pre-arranged on the host
    const float *bias;   // [Co]
    float *y;            // [B][Co][T][Fy]
    int Ci, Co, CoPad, T, Fi, FP, Fy;
    int s, os, oo;       // input column = s*m + coloff ; output column = os*m + oo
    int colpad;          // patch column c holds input freq c - colpad
    int tlo_off;         // patch row 0 holds source time ta + tlo_off
    int rows_extra;      // patch rows = (tb - ta + 1) + rows_extra
    int ntap;
    int rowoff[kMaxTaps];
    int coloff[kMaxTaps];
    int CC, nchunk, tiles_per_wg, St;
    int relu_lo, relu_hi;  // output channels in [relu_lo, relu_hi) get ReLU
};

template <int NT>
__global__ __launch_bounds__(256) void k_conv_igemm(ConvArgs a) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int P = a.T * a.FP;
    const int p0 = blockIdx.x * a.tiles_per_wg * 32;
    if (p0 >= P) return;
    const int p1 = min(P, p0 + a.tiles_per_wg * 32);
    const int ntile = (p1 - p0 + 31) >> 5;
    const int ta = p0 / a.FP, tb = (p1 - 1) / a.FP;
    const int R = tb - ta + 1 + a.rows_extra;
    const int St = a.St, Sc = R * St;
    const int CC = a.CC, CoPad = a.CoPad;
    float *wl = lds;                            // [ntap][CC][CoPad]  (first: keeps 16-B alignment for float4 copies)
    float *patch = lds + a.ntap * CC * CoPad;   // [CC][R][St]
    const int MT = CoPad >> 5, NCG = 4 / MT;
    const int mt = wave % MT, cg = wave / MT;
    const int half = lane >> 5, l31 = lane & 31;

    int lane_base[NT], pos_t[NT], pos_m[NT];
    bool tile_ok[NT], lane_ok[NT];
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const int tile = cg + i * NCG;
        tile_ok[i] = tile < ntile;
        const int p = p0 + tile * 32 + l31;
        lane_ok[i] = tile_ok[i] && p < p1;
        const int pc = lane_ok[i] ? p : (p1 - 1);
        const int t = pc / a.FP, m = pc - t * a.FP;
        pos_t[i] = t;
        pos_m[i] = m;
        lane_base[i] = (t - ta) * St + a.s * m + half * Sc;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;

    const int tlo = ta + a.tlo_off;
    const long xs_c = (long)a.T * a.Fi;  // channel stride in x
    const float *xb = a.x + (long)b * a.Ci * xs_c;
    const float *xpb = a.xprev ? a.xprev + (long)b * a.Ci * xs_c : nullptr;
    const int wslab = a.ntap * CC * CoPad;

    for (int ch = 0; ch < a.nchunk; ch++) {
        const int ci0 = ch * CC;
        __syncthreads();  // previous chunk fully consumed
        // ---- stage input patch (zero halo, causal history from xprev) ----
        for (int rid = wave; rid < CC * R; rid += 4) {
            const int c = rid / R, r = rid - c * R;
            const int ci = ci0 + c, ts = tlo + r;
            const float *srow = nullptr;
            if (ci < a.Ci) {
                if (ts >= 0 && ts < a.T) srow = xb + ci * xs_c + (long)ts * a.Fi;
                else if (ts < 0 && xpb && ts + a.T >= 0) srow = xpb + ci * xs_c + (long)(ts + a.T) * a.Fi;
            }
            float *drow = patch + c * Sc + r * St;
            for (int col = lane; col < St; col += 64) {
                const int fi = col - a.colpad;
                drow[col] = (srow && fi >= 0 && fi < a.Fi) ? srow[fi] : 0.0f;
            }
        }
        // ---- stage weight slab (contiguous copy) ----
        {
            const float4 *wsrc = reinterpret_cast<const float4 *>(a.w + (long)ch * wslab);
            float4 *wdst = reinterpret_cast<float4 *>(wl);
            for (int i = tid; i < (wslab >> 2); i += 256) wdst[i] = wsrc[i];
        }
        __syncthreads();
        // ---- MFMA over taps x channel pairs ----
        for (int tap = 0; tap < a.ntap; tap++) {
            const int toff = a.rowoff[tap] * St + a.coloff[tap];
            const float *wt = wl + tap * CC * CoPad + mt * 32 + l31 + half * CoPad;
            for (int kp = 0; kp < (CC >> 1); kp++) {
                const float av = wt[kp * 2 * CoPad];
                const int boff = toff + kp * 2 * Sc;
#pragma unroll
                for (int i = 0; i < NT; i++) {
                    if (tile_ok[i]) {
                        const float bv = patch[lane_base[i] + boff];
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---- epilogue: bias, ReLU, store [B][Co][T][Fy] ----
    const long ys_c = (long)a.T * a.Fy;
    float *yb = a.y + (long)b * a.Co * ys_c;
#pragma unroll
    for (int i = 0; i < NT; i++) {
        if (!lane_ok[i]) continue;
        float *yp = yb + (long)pos_t[i] * a.Fy + a.os * pos_m[i] + a.oo;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (co < a.Co) {
                float v = acc[i][r] + a.bias[co];
                if (co >= a.relu_lo && co < a.relu_hi) v = fmaxf(v, 0.0f);
                yp[co * ys_c] = v;
            }
        }
    }
}

}  // namespace se
