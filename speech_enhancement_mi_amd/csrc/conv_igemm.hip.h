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
#include "conv_args.h"

namespace se {

// block-wide sum of (s, q) over 256 threads -> one slab slot
__device__ __forceinline__ void conv_stats_store(const ConvArgs &a, float s, float q, float *red /*[8]*/, int b) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); q += __shfl_down(q, off, 64); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { red[wave * 2] = s; red[wave * 2 + 1] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float *o = a.stats + ((long)b * a.stats_nslot + a.stats_slot0 + blockIdx.x) * 2;
        o[0] = (red[0] + red[2]) + (red[4] + red[6]);
        o[1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
}

__device__ __forceinline__ float conv_act(float v, int act) {
    return act == 1 ? fmaxf(v, 0.0f) : (act == 2 ? (v > 0.0f ? v : expf(v) - 1.0f) : v);
}

// Epilogue shared by k_conv_igemm and k_conv_x6 (both leave 32x32 accumulator tiles: column = position on the lane,
// rows = GEMM rows in the 16 registers): bias, activation, optional gated-pair product, store [B][Cy][T][Fy], and the
// per-workgroup partial (sum, sum of squares) for the global layer norm that follows.
struct BlendStats { float my, iy, mu, iu; };  // mean / 1/std of ydec and of residualmask for this stream

template <int NT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs &a, const f32x16 (&acc)[NT], const bool (&lane_ok)[NT],
                                              const int (&pos_t)[NT], const int (&pos_m)[NT], int mt, int half, float *scratch, int b,
                                              const BlendStats *bs = nullptr) {
    const long ys_c = (long)a.T * a.Fy;
    float *yb = a.y + ((long)b * a.Cy + a.cy0) * ys_c;
    float ssum = 0.0f, ssq = 0.0f;
    if (a.blend && bs) {  // fused decoder skip gate: 8 (residualmask, residual) row pairs per lane
        float pb[16], cnw[8], cnb[8], cmw[8], cmb[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int row = mt * 32 + ((2 * q) & 3) + 8 * ((2 * q) >> 2) + 4 * half;  // even
            const int c = min(row >> 1, a.Cy - 1);
            pb[2 * q] = a.bias[min(row, a.Co - 1)];
            pb[2 * q + 1] = a.bias[min(row + 1, a.Co - 1)];
            cnw[q] = a.bl_nw[c]; cnb[q] = a.bl_nb[c]; cmw[q] = a.bl_mnw[c]; cmb[q] = a.bl_mnb[c];
        }
        const long yd_c = (long)a.T * a.bl_Fo;
        const float *ydb = a.bl_ydec + (long)b * a.Cy * yd_c;
#pragma unroll
        for (int i = 0; i < NT; i++) {
            if (!lane_ok[i]) continue;
            const int f = a.os * pos_m[i] + a.oo;
            const bool has_y = f < a.bl_Fo;  // the transposed convolution gives 2*Fi - 1 bins: pad with zeros (CRN.py:389-392)
            const float *ydp = ydb + (long)pos_t[i] * a.bl_Fo + min(f, a.bl_Fo - 1);
            float yv[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {  // all loads of the tile before its first store
                const int row = mt * 32 + ((2 * q) & 3) + 8 * ((2 * q) >> 2) + 4 * half;
                yv[q] = ydp[(long)min(row >> 1, a.Cy - 1) * yd_c];
            }
#pragma unroll
            for (int q = 0; q < 8; q++) asm volatile("" : "+v"(yv[q]));
            float *yp = yb + (long)pos_t[i] * a.Fy + f;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int row = mt * 32 + ((2 * q) & 3) + 8 * ((2 * q) >> 2) + 4 * half;
                if (row + 1 < a.Co) {
                    const float u = acc[i][2 * q] + pb[2 * q];
                    const float v = conv_act(acc[i][2 * q + 1] + pb[2 * q + 1], a.act);
                    const float yn = has_y ? (yv[q] - bs->my) * bs->iy * cnw[q] + cnb[q] : 0.0f;
                    const float un = (u - bs->mu) * bs->iu * cmw[q] + cmb[q];
                    const float m = 1.0f / (1.0f + expf(-un));
                    yp[(row >> 1) * ys_c] = m * v + (1.0f - m) * yn;
                }
            }
        }
        return;
    }
    // the 16 biases of this lane's rows are fetched ONCE, ahead of all stores: a bias load between two stores cannot be
    // hoisted by the compiler (y may alias bias for all it knows) and would cost one L1 round trip per stored element
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        bv[r] = a.bias[min(co, a.Co - 1)];
    }
#pragma unroll
    for (int r = 0; r < 16; r++) asm volatile("" : "+v"(bv[r]));
#pragma unroll
    for (int i = 0; i < NT; i++) {
        if (!lane_ok[i]) continue;
        float *yp = yb + (long)pos_t[i] * a.Fy + a.os * pos_m[i] + a.oo;
        if (a.gate_pairs) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;  // even
                if (row + 1 < a.Co) {
                    const float tr = acc[i][r] + bv[r], gt = acc[i][r + 1] + bv[r + 1];
                    const float v = tr * (1.0f / (1.0f + expf(-gt)));
                    yp[(row >> 1) * ys_c] = v;
                    ssum += v; ssq += v * v;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < a.Co) {
                    float v = acc[i][r] + bv[r];
                    if (co >= a.relu_lo && co < a.relu_hi) v = conv_act(v, a.act);
                    if (a.par_rows) {
                        const int par = co & 1;
                        if (a.os * pos_m[i] + a.oo + par < a.Fy) {  // the odd parity has one column less
                            yp[(co >> 1) * ys_c + par] = v;
                            ssum += v; ssq += v * v;
                        }
                    } else {
                        if (a.y) yp[co * ys_c] = v;  // y == nullptr: statistics-only pass
                        if (co >= a.stats_lo && co < a.stats_hi) { ssum += v; ssq += v * v; }
                    }
                }
            }
        }
    }
    if (a.stats) conv_stats_store(a, ssum, ssq, scratch, b);
}


// Issues (does not wait for) the global loads of chunk `ch`: this thread's patch elements and weight slots.
__device__ __forceinline__ void conv_issue_loads(const ConvArgs &a, int ch, int tid, int CC, int Sc, int npatch, long xs_c,
                                                 const float *xb, const float *xpb, const int (&goff)[kPatchPerThread],
                                                 unsigned okmask, unsigned histmask, int wslab, int n4,
                                                 float (&pv)[kPatchPerThread], f32x4 (&wv)[kWeightPerThread]) {
    const int cbase = ch * CC;
    const int cmaxe = (a.Ci - cbase) * Sc;  // elements of channels >= Ci are zero
    const long chan_off = (long)cbase * xs_c;
    // the last chunk may be partial: clamp the base so every address stays inside x
    const long safe_off = (long)min(cbase, max(a.Ci - CC, 0)) * xs_c;
#pragma unroll
    for (int k = 0; k < kPatchPerThread; k++) {
        const int e = tid + 256 * k;
        const bool in = e < min(cmaxe, npatch);
        const float *base = ((histmask >> k) & 1u) ? xpb : xb;
        pv[k] = base[(in ? chan_off : safe_off) + goff[k]];  // raw: the zero mask is applied at the LDS write,
                                                              // so no load is waited for here
    }
    const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(a.w + (long)ch * wslab);
#pragma unroll
    for (int k = 0; k < kWeightPerThread; k++) wv[k] = wsrc[min(tid + 256 * k, n4 - 1)];
}

// NTAP = number of taps of this tap set (15 encoder, 9 / 6 transposed even / odd, 1 for 1x1): a template
// parameter so the tap loop unrolls and the per-tap LDS offsets live in SGPRs; NT = column tiles per wave.
//
// Staging is a register-prefetch pipeline: every thread owns up to 16 fixed patch elements and 8 float4 weight
// slots of a chunk.  Their global offsets never change from chunk to chunk except for the channel base, so the
// gather plan (offset, zero-halo / history flags) is computed once per workgroup; while the MFMAs of chunk k run
// out of LDS, the loads of chunk k+1 are already in flight into registers and are written to LDS after the
// barrier.  Global-memory latency is therefore hidden inside one workgroup instead of relying on neighbours.
// The MFMA loop is branch-free: column tiles beyond the workgroup's range read a clamped address and their
// accumulators are simply never stored.
template <int NTAP, int NT>
__global__ __launch_bounds__(256, 2) void k_conv_igemm(ConvArgs a) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int P = a.T * a.FP;
    const int p0 = blockIdx.x * a.tiles_per_wg * 32;
    if (p0 >= P) return;
    const int p1 = min(P, p0 + a.tiles_per_wg * 32);
    const int ta = p0 / a.FP, tb = (p1 - 1) / a.FP;
    const int RT = tb - ta + 1;
    const int R = a.grouped ? a.ngroup * RT : RT + (a.ngroup - 1) * a.dil;
    const int St = a.St, Sc = R * St;
    const int CC = a.CC, CoPad = a.CoPad;
    float *wl = lds;                          // [NTAP][CC][CoPad]  (first: keeps 16-B alignment for float4 copies)
    float *patch = lds + NTAP * CC * CoPad;   // [CC][R][St]
    const int MT = CoPad >> 5, NCG = 4 / MT;
    const int mt = wave % MT, cg = wave / MT;
    const int half = lane >> 5, l31 = lane & 31;

    int lane_base[NT], pos_t[NT], pos_m[NT];
    bool lane_ok[NT];
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const int p = p0 + (cg + i * NCG) * 32 + l31;
        lane_ok[i] = p < p1;
        const int pc = lane_ok[i] ? p : (p1 - 1);
        const int t = pc / a.FP, m = pc - t * a.FP;
        pos_t[i] = t;
        pos_m[i] = m;
        lane_base[i] = (t - ta) * St + a.s * m + half * Sc;
    }
    int toff[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; t++) toff[t] = a.rowgrp[t] * (a.grouped ? RT : a.dil) * St + a.coloff[t];

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;

    const long xs_c = (long)a.T * a.Fi;  // channel stride in x
    const float *xb = a.x + (long)b * a.Ci * xs_c;
    const float *xpb = a.xprev ? a.xprev + (long)b * a.Ci * xs_c : xb;
    const int wslab = NTAP * CC * CoPad, n4 = wslab >> 2;
    const int npatch = CC * Sc;
    const int hk = CC >> 1;

    // ---- gather plan: element e = tid + 256*k of the chunk patch [CC][R][St] ----
    int goff[kPatchPerThread];
    unsigned okmask = 0, histmask = 0;
    {
        const int NGp = a.grouped ? a.ngroup : 1, RTp = a.grouped ? RT : R;
#pragma unroll
        for (int k = 0; k < kPatchPerThread; k++) {
            const int e = min(tid + 256 * k, npatch - 1);
            const int row = e / St, col = e - row * St;
            const int c = row / R, r = row - c * R;
            const int g = NGp > 1 ? r / RTp : 0, j = r - g * RTp;
            const int ts = ta + a.tlo_off + g * a.dil + j;
            const int fi = col - a.colpad;
            const bool hist = ts < 0;
            const bool ok = fi >= 0 && fi < a.Fi && (hist ? (a.xprev != nullptr && ts + a.T >= 0) : ts < a.T);
            const int tsc = min(max(hist ? ts + a.T : ts, 0), a.T - 1), fic = min(max(fi, 0), a.Fi - 1);
            goff[k] = c * (int)xs_c + tsc * a.Fi + fic;
            okmask |= (ok ? 1u : 0u) << k;
            histmask |= (hist ? 1u : 0u) << k;
        }
    }
    float pv[kPatchPerThread];
    f32x4 wv[kWeightPerThread];
    conv_issue_loads(a, 0, tid, CC, Sc, npatch, xs_c, xb, xpb, goff, okmask, histmask, wslab, n4, pv, wv);

    for (int ch = 0; ch < a.nchunk; ch++) {
        __syncthreads();  // previous chunk fully consumed
        {
            const int cmaxe = (a.Ci - ch * CC) * Sc;  // elements of channels >= Ci are zero
#pragma unroll
            for (int k = 0; k < kPatchPerThread; k++) {
                const int e = tid + 256 * k;
                if (e < npatch) patch[e] = (e < cmaxe && ((okmask >> k) & 1u)) ? pv[k] : 0.0f;
            }
        }
#pragma unroll
        for (int k = 0; k < kWeightPerThread; k++)
            if (tid + 256 * k < n4) reinterpret_cast<f32x4 *>(wl)[tid + 256 * k] = wv[k];
        __syncthreads();
        if (ch + 1 < a.nchunk)  // in flight during the MFMAs below
            conv_issue_loads(a, ch + 1, tid, CC, Sc, npatch, xs_c, xb, xpb, goff, okmask, histmask, wslab, n4, pv, wv);
        // software pipeline, one (tap, channel pair) step deep: the LDS reads of step s+1 are issued before
        // the NT MFMAs of step s (256 matrix-pipe cycles cover the ~100-cycle LDS latency).
        int wofs = mt * 32 + l31 + half * CoPad;  // A fragment offset inside wl (floats)
        int lb[NT];
#pragma unroll
        for (int i = 0; i < NT; i++) lb[i] = lane_base[i];
        float av = wl[wofs], bv[NT];
#pragma unroll
        for (int i = 0; i < NT; i++) bv[i] = patch[lb[i] + toff[0]];
        const int tapA = CC * CoPad;
        for (int kp = 0; kp < hk; kp++) {
            // opaque to the optimiser: keeps the 5*NTAP fragment addresses from being hoisted out of the loop
            // (75 live address registers spill); one v_add per LDS read is free next to a 64-cycle MFMA.
            asm volatile("" : "+v"(wofs));
#pragma unroll
            for (int i = 0; i < NT; i++) asm volatile("" : "+v"(lb[i]));
            const int ka = kp * 2 * CoPad, kb = kp * 2 * Sc;
            const int kpn = kp + 1 < hk ? kp + 1 : 0;  // last step prefetches a valid (unused) address
            const int kan = kpn * 2 * CoPad, kbn = kpn * 2 * Sc;
#pragma unroll
            for (int tap = 0; tap < NTAP; tap++) {
                float an, bn[NT];
                if (tap + 1 < NTAP) {
                    an = wl[wofs + ka + (tap + 1) * tapA];
#pragma unroll
                    for (int i = 0; i < NT; i++) bn[i] = patch[lb[i] + kb + toff[tap + 1]];
                } else {
                    an = wl[wofs + kan];
#pragma unroll
                    for (int i = 0; i < NT; i++) bn[i] = patch[lb[i] + kbn + toff[0]];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NT; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[i], acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                av = an;
#pragma unroll
                for (int i = 0; i < NT; i++) bv[i] = bn[i];
            }
        }
    }
    __syncthreads();
    conv_epilogue<NT>(a, acc, lane_ok, pos_t, pos_m, mt, half, lds, b);  // lds is free again: all waves are past the MFMA loop
}

// Convolutions with <= 8 output channels (the last decoder block, 16 -> 2 channels, and the 5 -> 5 channel 5x5
// preconv blocks of CRN_ELU: a 32-row MFMA tile would be 84-94 % padding) run on the vector ALU, one thread per
// output position, reading the input straight from L2/HBM: consecutive lanes read consecutive frequencies
// (coalesced) and all NTAP loads of a channel are independent, so the memory system stays full; an LDS patch (as in
// k_conv_igemm) would re-stage a time-tap halo several times larger than the rows a workgroup produces.
// Weights [tap][ci][4*W4] sit in LDS and are read as broadcast float4.
// The 5x5 frequency-dilated pre-conv blocks of CRN_ELU / the student (CRN_ELU.py:335-340: CO -> CO channels, taps (kf, kt),
// time rows t-4..t, columns f + (kf - 2) fd) on the vector ALU, TWO consecutive time rows per thread: the outputs at t and
// t + 1 share four of their five tap rows, so a thread loads 6 rows x 5 columns per channel for both (150 loads instead of
// 250), every weight read from LDS feeds both outputs (half the broadcast reads of k_conv_small) and exactly CO accumulators
// per output are kept (625 FMAs per output instead of 1000).  Weights [25][Ci][8] and the fused gated pair as k_conv_small.
template <int CO>
__global__ __launch_bounds__(256) void k_preconv_tb(ConvArgs a) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, b = blockIdx.y;
    if ((int)blockIdx.x * 256 >= ((a.T + 1) >> 1) * a.FP) {  // the launch grid is sized for one output per thread: the second half of the
        if (a.stats && tid == 0) {                            // workgroups has no row pair left and only reports empty statistics
            float *o = a.stats + ((long)b * a.stats_nslot + a.stats_slot0 + blockIdx.x) * 2;
            o[0] = 0.0f; o[1] = 0.0f;
        }
        return;
    }
    const int nw = 25 * a.Ci * 8;
    for (int i = tid; i < nw; i += 256) lds[i] = a.w[i];
    float *gl = lds + nw;
    const int ng = 2 * CO * CO + 2 * CO;
    for (int i = tid; i < ng; i += 256) gl[i] = a.gatew[i];
    __syncthreads();
    const int F = a.FP, TP = (a.T + 1) >> 1;  // row pairs
    const int p = blockIdx.x * 256 + tid;
    const bool live = p < TP * F;
    const int pc = live ? p : TP * F - 1;
    const int tp = pc / F, f = pc - tp * F, t0 = 2 * tp;
    const bool two = t0 + 1 < a.T;
    const long xs_c = (long)a.T * a.Fi;
    const float *xb = a.x + (long)b * a.Ci * xs_c;
    const float *xpb = a.xprev ? a.xprev + (long)b * a.Ci * xs_c : nullptr;
    // rows t0-4 .. t0+1 (history rows from the previous slot), columns f + coloff[kf * 5] - colpad
    const float *rowp[6];
    bool rok[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const int ts = t0 - 4 + j;
        const bool hist = ts < 0;
        rok[j] = hist ? (xpb != nullptr && ts + a.T >= 0) : ts < a.T;
        const int tsc = min(max(hist ? ts + a.T : ts, 0), a.T - 1);
        rowp[j] = (hist && xpb ? xpb : xb) + (long)tsc * a.Fi;
    }
    int col[5];
    bool cok[5];
#pragma unroll
    for (int kf = 0; kf < 5; kf++) {
        const int fi = f + a.coloff[kf * 5] - a.colpad;
        cok[kf] = fi >= 0 && fi < a.Fi;
        col[kf] = min(max(fi, 0), a.Fi - 1);
    }
    float acc0[CO], acc1[CO];
#pragma unroll
    for (int j = 0; j < CO; j++) acc0[j] = acc1[j] = 0.0f;
    for (int c = 0; c < a.Ci; c++) {
#pragma unroll
        for (int kf = 0; kf < 5; kf++) {
            float v[6];
#pragma unroll
            for (int j = 0; j < 6; j++) v[j] = rowp[j][c * xs_c + col[kf]];
#pragma unroll
            for (int j = 0; j < 6; j++) v[j] = (rok[j] && cok[kf]) ? v[j] : 0.0f;
#pragma unroll
            for (int kt = 0; kt < 5; kt++) {
                const float *w = lds + ((kf * 5 + kt) * a.Ci + c) * 8;
                const float4 w4 = *reinterpret_cast<const float4 *>(w);
                float wv[8] = {w4.x, w4.y, w4.z, w4.w, 0, 0, 0, 0};
                if (CO > 4) {
                    const float4 w5 = *reinterpret_cast<const float4 *>(w + 4);
                    wv[4] = w5.x; wv[5] = w5.y; wv[6] = w5.z; wv[7] = w5.w;
                }
#pragma unroll
                for (int j = 0; j < CO; j++) { acc0[j] += wv[j] * v[kt]; acc1[j] += wv[j] * v[kt + 1]; }
            }
        }
    }
    const long ys_c = (long)a.T * a.Fy;
    float ssum = 0.0f, ssq = 0.0f;
#pragma unroll
    for (int o = 0; o < 2; o++) {
        float val[CO], outv[CO];
#pragma unroll
        for (int co = 0; co < CO; co++) val[co] = conv_act((o ? acc1[co] : acc0[co]) + a.bias[co], a.act);
#pragma unroll
        for (int co = 0; co < CO; co++) {  // out = conv_trans(v) * sigmoid(conv_gated(v))  (CRN_ELU.py:240)
            float tr = gl[2 * CO * CO + co], gt = gl[2 * CO * CO + CO + co];
#pragma unroll
            for (int j = 0; j < CO; j++) { tr += gl[co * CO + j] * val[j]; gt += gl[CO * CO + co * CO + j] * val[j]; }
            outv[co] = tr * (1.0f / (1.0f + expf(-gt)));
        }
        if (live && (o == 0 || two)) {
            float *yp = a.y + ((long)b * a.Cy + a.cy0) * ys_c + (long)(t0 + o) * a.Fy + f + a.oo;
#pragma unroll
            for (int co = 0; co < CO; co++) {
                yp[co * ys_c] = outv[co];
                if (co >= a.stats_lo && co < a.stats_hi) { ssum += outv[co]; ssq += outv[co] * outv[co]; }
            }
        }
    }
    __syncthreads();  // weights no longer needed: lds doubles as the reduction scratch
    if (a.stats) conv_stats_store(a, ssum, ssq, lds, b);
}

template <int NTAP, int W4>
__global__ __launch_bounds__(256) void k_conv_small(ConvArgs a) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int P = a.T * a.FP;
    constexpr int CW = 4 * W4;
    const int nw = NTAP * a.Ci * CW;  // host lays the weights out as ONE chunk: [NTAP][Ci][CW]
    for (int i = tid; i < nw; i += 256) lds[i] = a.w[i];
    float *gl = lds + nw;
    const int ng = a.gatew ? 2 * a.Co * a.Co + 2 * a.Co : 0;
    for (int i = tid; i < ng; i += 256) gl[i] = a.gatew[i];
    __syncthreads();
    const int p = blockIdx.x * 256 + tid;
    const bool live = p < P;
    const int pc = live ? p : P - 1;
    const int t = pc / a.FP, m = pc - t * a.FP;
    const long xs_c = (long)a.T * a.Fi;
    const float *xb = a.x + (long)b * a.Ci * xs_c;
    const float *xpb = a.xprev ? a.xprev + (long)b * a.Ci * xs_c : xb;
    const float *src[NTAP];
    unsigned okbits = 0;
#pragma unroll
    for (int k = 0; k < NTAP; k++) {
        const int ts = t + a.tlo_off + a.rowgrp[k] * a.dil;
        const int fi = a.s * m + a.coloff[k] - a.colpad;
        const bool hist = ts < 0;
        const bool ok = fi >= 0 && fi < a.Fi && (hist ? (a.xprev != nullptr && ts + a.T >= 0) : ts < a.T);
        okbits |= (ok ? 1u : 0u) << k;
        const int tsc = min(max(hist ? ts + a.T : ts, 0), a.T - 1), fic = min(max(fi, 0), a.Fi - 1);
        src[k] = (hist ? xpb : xb) + (long)tsc * a.Fi + fic;
    }
    float acc[CW];
#pragma unroll
    for (int j = 0; j < CW; j++) acc[j] = 0.0f;
    for (int c = 0; c < a.Ci; c++) {
        float v[NTAP];
#pragma unroll
        for (int k = 0; k < NTAP; k++) v[k] = src[k][c * xs_c];
#pragma unroll
        for (int k = 0; k < NTAP; k++) {
            const float x = ((okbits >> k) & 1u) ? v[k] : 0.0f;
#pragma unroll
            for (int q = 0; q < W4; q++) {
                const float4 w = *reinterpret_cast<const float4 *>(lds + (k * a.Ci + c) * CW + 4 * q);
                acc[4 * q] += w.x * x; acc[4 * q + 1] += w.y * x; acc[4 * q + 2] += w.z * x; acc[4 * q + 3] += w.w * x;
            }
        }
    }
    const long ys_c = (long)a.T * a.Fy;
    float *yp = a.y + ((long)b * a.Cy + a.cy0) * ys_c + (long)t * a.Fy + a.os * m + a.oo;
    float ssum = 0.0f, ssq = 0.0f;
    float val[CW];
#pragma unroll
    for (int co = 0; co < CW; co++) {
        float v = co < a.Co ? acc[co] + a.bias[co] : 0.0f;
        if (co >= a.relu_lo && co < a.relu_hi) v = conv_act(v, a.act);
        val[co] = v;
    }
    if (a.gatew) {  // out = conv_trans(v) * sigmoid(conv_gated(v))  (CRN_ELU.py:240), channel mixing in registers
        const int C = a.Co;
        float outv[CW];
#pragma unroll
        for (int co = 0; co < CW; co++) {
            float tr = 0.0f, gt = 0.0f;
            if (co < C) {
                tr = gl[2 * C * C + co];
                gt = gl[2 * C * C + C + co];
#pragma unroll
                for (int j = 0; j < CW; j++)
                    if (j < C) { tr += gl[co * C + j] * val[j]; gt += gl[C * C + co * C + j] * val[j]; }
            }
            outv[co] = tr * (1.0f / (1.0f + expf(-gt)));
        }
#pragma unroll
        for (int co = 0; co < CW; co++) val[co] = outv[co];
    }
#pragma unroll
    for (int co = 0; co < CW; co++)
        if (co < a.Co && live) {
            yp[co * ys_c] = val[co];
            if (co >= a.stats_lo && co < a.stats_hi) { ssum += val[co]; ssq += val[co] * val[co]; }
        }
    __syncthreads();  // gl / weights no longer needed: lds doubles as the reduction scratch
    if (a.stats) conv_stats_store(a, ssum, ssq, lds, b);
}

}  // namespace se
