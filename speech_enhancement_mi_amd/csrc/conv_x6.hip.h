// conv_x6.hip.h - the implicit-GEMM convolution of conv_igemm.hip.h on the bf16 matrix cores with fp32 accuracy.
//
// Same GEMM view, tiling, gather plan, epilogue and statistics as k_conv_igemm; what changes is the arithmetic:
// every fp32 operand is split into three bf16 planes (x = hi + mid + lo, 24 mantissa bits) and each product is
// formed from the six leading cross terms with v_mfma_f32_32x32x16_bf16 (see gemm.hip.h, k_gemm_bf16x6: measured
// error 2.6e-8 vs 2.6e-7 for a k-ordered fp32 chain).  6 x 32 matrix-pipe cycles per 16-deep K step replace
// 8 x 64 cycles of v_mfma_f32_32x32x2_f32.
//   K step  = 2 (tap, channel-octet) entries x 8 input channels: lane half h of the MFMA operand selects the entry,
//             j the channel inside the octet.  A chunk stages CO octets (8*CO channels); its NTAP*CO entries are
//             walked tap-major, so a 1x1 convolution (NTAP = 1) contracts 16 real channels per MFMA instead of
//             8 channels + 8 zeros, and needs CO times fewer chunk barriers.
//   patch   = LDS [plane 3][octet CO][row][col][8 ch] bf16: one 16-B ds_read_b128 per B fragment at
//             lane_base + entry offset
//   weights = pre-split on the host as [chunk][tap pair][plane][Cout tile][32 co][16 k]: one coalesced 1-KB
//             global load per A fragment straight to registers (prefetched one pair ahead; no LDS weight slab,
//             so the LDS holds three patch planes instead)
//   staging = each thread owns up to 4 (position, octet) items x 8 channels; fp32 values are prefetched into registers
//             during the previous chunk's MFMAs, split and written as three 16-B LDS stores per position.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_igemm.hip.h"
#include "split_bf16.h"

namespace se {

__device__ __forceinline__ uint4 pack_bf16x8(const __bf16 (&v)[8]) {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = v[i];
    return __builtin_bit_cast(uint4, t);
}

#ifdef SE_X6_TRACE
__device__ unsigned long long g_x6_trace[16];
#define X6T(i) do { const unsigned long long _n = __builtin_readcyclecounter(); if (tid == 0) tr[i] += _n - tlast; tlast = _n; } while (0)
#else
#define X6T(i)
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// PL = operand planes: 3 = fp32-accurate bf16x6 (six cross terms per product); 2 = bf16x3 (hi, mid planes; three cross terms
// hi*hi + hi*mid + mid*hi: 16 mantissa bits per operand, measured 4e-6 relative on a K = 1664 GEMM, se_config.precision = 2);
// 1 = plain fp16 operands with fp32
// accumulation (se_config.precision = 1: the fp16 inference mode of BASELINE config 5, 6x fewer MFMAs, a third of the LDS
// traffic; operands rounded to 11 mantissa bits)
template <int NTAP, int NT, int CO, int PL>
__global__ __launch_bounds__(256, NT <= 2 ? 3 : 2) void k_conv_x6(ConvX6Args xa) {
    extern __shared__ __align__(16) uint4 planes[];  // [PL][CO][Npos]
    const ConvArgs &a = xa.c;
    constexpr int NPAIR = (NTAP * CO + 1) / 2;  // K steps per chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    BlendStats bs{0.f, 1.f, 0.f, 1.f};
    if (NTAP == 1 && a.blend) {  // fused decoder skip gate: this stream's norm statistics (LDS is still free)
        float *sm = reinterpret_cast<float *>(planes);
        slab_mean_inv(a.bl_sy, b, sm, bs.my, bs.iy);
        slab_mean_inv(a.bl_su, b, sm + 2, bs.mu, bs.iu);
        __syncthreads();
    }
    const int P = a.T * a.FP;
    const int p0 = blockIdx.x * a.tiles_per_wg * 32;
    if (p0 >= P) return;
    const int p1 = min(P, p0 + a.tiles_per_wg * 32);
    const int ta = p0 / a.FP, tb = (p1 - 1) / a.FP;
    const int RT = tb - ta + 1;
    const int R = a.grouped ? a.ngroup * RT : RT + (a.ngroup - 1) * a.dil;
    const int St = a.St, Npos = R * St;
    const int MT = a.CoPad >> 5, NCG = 4 / MT;
    const int mt = wave % MT, cg = wave / MT;
    const int half = lane >> 5, l31 = lane & 31;

    int lane_base[NT];  // (the epilogue's position bookkeeping is recomputed after the loop: it would only hold registers)
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const int pc = min(p0 + (cg + i * NCG) * 32 + l31, p1 - 1);
        const int t = pc / a.FP, m = pc - t * a.FP;
        lane_base[i] = (t - ta) * St + a.s * m;
    }
    int toffL[NPAIR];  // per lane half: LDS offset of entry 2*step + half = (tap, octet), tap-major
#pragma unroll
    for (int pr = 0; pr < NPAIR; pr++) {
        const int en = min(2 * pr + half, NTAP * CO - 1);  // a padded (zero-weight) entry reads any valid position
        const int tp = en / CO, oc = en - tp * CO;
        toffL[pr] = oc * Npos + a.rowgrp[tp] * (a.grouped ? RT : a.dil) * St + a.coloff[tp];
    }
    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;

    const long xs_c = (long)a.T * a.Fi;
    const float *xb = a.x + (long)b * a.Ci * xs_c;
    const float *xpb = a.xprev ? a.xprev + (long)b * a.Ci * xs_c : xb;

    // ---- gather plan over items it = tid + 256*k = octet * Npos + patch position ----
    int goff[kX6PosPerThread];
    unsigned okmask = 0, histmask = 0, octs = 0;  // octs: 2 bits per item = its channel octet inside the chunk
    {
        const int NGp = a.grouped ? a.ngroup : 1, RTp = a.grouped ? RT : R;
#pragma unroll
        for (int k = 0; k < kX6PosPerThread; k++) {
            const int item = min(tid + 256 * k, CO * Npos - 1);
            const int pe = CO > 1 ? item % Npos : item;
            const int r = pe / St, col = pe - r * St;
            const int g = NGp > 1 ? r / RTp : 0, j = r - g * RTp;
            const int ts = ta + a.tlo_off + g * a.dil + j;
            const int fi = col - a.colpad;
            const bool hist = ts < 0;
            const bool ok = fi >= 0 && fi < a.Fi && (hist ? (a.xprev != nullptr && ts + a.T >= 0) : ts < a.T);
            const int tsc = min(max(hist ? ts + a.T : ts, 0), a.T - 1), fic = min(max(fi, 0), a.Fi - 1);
            goff[k] = tsc * a.Fi + fic;
            octs |= (unsigned)(CO > 1 ? item / Npos : 0) << (2 * k);
            okmask |= (ok ? 1u : 0u) << k;
            histmask |= (hist ? 1u : 0u) << k;
        }
    }
    float pv[kX6PosPerThread][8];
    auto issue_loads = [&](int ch) {
#pragma unroll
        for (int k = 0; k < kX6PosPerThread; k++) {
            const float *base = ((histmask >> k) & 1u) ? xpb : xb;
            const int cbase = ch * 8 * CO + (CO > 1 ? (int)((octs >> (2 * k)) & 3u) * 8 : 0);
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const int ci = min(cbase + c, a.Ci - 1);
                pv[k][c] = base[(long)ci * xs_c + goff[k]];  // raw; masked at the LDS write
            }
        }
    };
    issue_loads(0);
    const uint4 *wxw = xa.wx + (long)mt * 64 + l31 * 2 + half;  // + (((ch*NPAIR + pr)*3 + plane)*MT) * 64
    const long wx_plane = (long)MT * 64, wx_pair = PL * wx_plane, wx_chunk = NPAIR * wx_pair;

#ifdef SE_X6_TRACE
    unsigned long long tr[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    for (int ch = 0; ch < a.nchunk; ch++) {
        X6T(0);
        // The weight fragments of the chunk's first K step are requested before the staging phase where registers allow
        // (<= 9 taps), which hides their latency: requested after the second barrier they cost 6-13 k exposed cycles per
        // workgroup (phase trace); the 15-tap instances are at the register limit and spill when the request moves up.
        constexpr bool kEarlyA = NTAP <= 9;
        const uint4 *wc = wxw + ch * wx_chunk;
        uint4 fa_n[PL];
        if (kEarlyA) {
#pragma unroll
            for (int p = 0; p < PL; p++) fa_n[p] = wc[p * wx_plane];
        }
        __syncthreads();  // previous chunk fully consumed
        X6T(1);
#ifdef SE_X6_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        X6T(2);
#endif
#pragma unroll
        for (int k = 0; k < kX6PosPerThread; k++) {
            const int item = tid + 256 * k;
            if (item < CO * Npos) {
                const bool ok = (okmask >> k) & 1u;
                const int cbase = ch * 8 * CO + (CO > 1 ? (int)((octs >> (2 * k)) & 3u) * 8 : 0);
                if (PL >= 2) {
                    __bf16 h[8], m[8], l[8];
#pragma unroll
                    for (int c = 0; c < 8; c++) split3((ok && cbase + c < a.Ci) ? pv[k][c] : 0.0f, h[c], m[c], l[c]);
                    planes[item] = pack_bf16x8(h);
                    planes[CO * Npos + item] = pack_bf16x8(m);
                    if (PL == 3) planes[2 * CO * Npos + item] = pack_bf16x8(l);
                } else {
                    f16x8 hv;
#pragma unroll
                    for (int c = 0; c < 8; c++) hv[c] = (_Float16)((ok && cbase + c < a.Ci) ? pv[k][c] : 0.0f);
                    planes[item] = __builtin_bit_cast(uint4, hv);
                }
            }
        }
        X6T(3);
        __syncthreads();
        X6T(4);
        if (!kEarlyA) {
#pragma unroll
            for (int p = 0; p < PL; p++) fa_n[p] = wc[p * wx_plane];
        }
#ifdef SE_X6_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        X6T(5);
#endif
#pragma unroll
        for (int pr = 0; pr < NPAIR; pr++) {
            uint4 fa[PL];
#pragma unroll
            for (int p = 0; p < PL; p++) fa[p] = fa_n[p];
            if (pr + 1 < NPAIR) {
#pragma unroll
                for (int p = 0; p < PL; p++) fa_n[p] = wc[(pr + 1) * wx_pair + p * wx_plane];
            }
            // vmcnt retires in order: the next chunk's 32 staging loads are issued only after the LAST weight-fragment
            // load of this chunk, otherwise the first in-loop fragment wait would also wait for all of them
            if (pr == (NPAIR >= 2 ? NPAIR - 2 : 0) && ch + 1 < a.nchunk) issue_loads(ch + 1);
#pragma unroll
            for (int i = 0; i < NT; i++) {
                const int pos = lane_base[i] + toffL[pr];
                f32x16 c = acc[i];
                if (PL >= 2) {
                    const bf16x8 a0 = __builtin_bit_cast(bf16x8, fa[0]), a1 = __builtin_bit_cast(bf16x8, fa[PL > 1 ? 1 : 0]),
                                 a2 = __builtin_bit_cast(bf16x8, fa[PL > 2 ? 2 : 0]);
                    const bf16x8 b0 = __builtin_bit_cast(bf16x8, planes[pos]);
                    const bf16x8 b1 = __builtin_bit_cast(bf16x8, planes[CO * Npos + pos]);
                    if (PL == 3) {
                        const bf16x8 b2 = __builtin_bit_cast(bf16x8, planes[2 * CO * Npos + pos]);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c, 0, 0, 0);  // mid*mid
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, c, 0, 0, 0);  // hi*lo
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, c, 0, 0, 0);  // lo*hi
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c, 0, 0, 0);  // hi*mid
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c, 0, 0, 0);  // mid*hi
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c, 0, 0, 0);  // hi*hi
                } else {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[0]), __builtin_bit_cast(f16x8, planes[pos]), c, 0, 0, 0);
                }
                acc[i] = c;
            }
        }
    }
    X6T(0);
    __syncthreads();
    X6T(1);
#ifdef SE_X6_TRACE
    {
        int pos_t[NT], pos_m[NT];
        bool lane_ok[NT];
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int p = p0 + (cg + i * NCG) * 32 + l31;
            lane_ok[i] = p < p1;
            const int pc = lane_ok[i] ? p : (p1 - 1);
            pos_t[i] = pc / a.FP;
            pos_m[i] = pc - pos_t[i] * a.FP;
        }
        ConvArgs a2 = a;
        a2.stats = nullptr;
        conv_epilogue<NT>(a2, acc, lane_ok, pos_t, pos_m, mt, half, reinterpret_cast<float *>(planes), b, NTAP == 1 ? &bs : nullptr);
        X6T(6);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        X6T(7);
        if (a.stats) conv_stats_store(a, 0.f, 0.f, reinterpret_cast<float *>(planes), b);
        X6T(8);
    }
    if (tid == 0 && xa.trace_slot >= 0) {
        for (int i = 0; i < 9; i++) atomicAdd(&g_x6_trace[i], tr[i]);
        atomicAdd(&g_x6_trace[15], 1ull);
    }
#else
    int pos_t[NT], pos_m[NT];
    bool lane_ok[NT];
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const int p = p0 + (cg + i * NCG) * 32 + l31;
        lane_ok[i] = p < p1;
        const int pc = lane_ok[i] ? p : (p1 - 1);
        pos_t[i] = pc / a.FP;
        pos_m[i] = pc - pos_t[i] * a.FP;
    }
    conv_epilogue<NT>(a, acc, lane_ok, pos_t, pos_m, mt, half, reinterpret_cast<float *>(planes), b, NTAP == 1 ? &bs : nullptr);
#endif
}

}  // namespace se
