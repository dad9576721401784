// conv_p.hip.h - second-generation implicit-GEMM convolution: MFMA operands live in HBM ALREADY SPLIT into bf16 planes.
//
// Same GEMM view as conv_x6.hip.h (M = Cout rows, N = output positions of one stream, K = taps x Cin walked in chunks of CO
// channel octets; a K step = two (tap, octet) entries x 8 channels; split-bf16 products hi*hi + hi*mid + mid*hi [+ hi*lo +
// lo*hi + mid*mid]), but the data path around the MFMA loop is different:
//
//   input   "P layout"  P[b][octet][plane][t][f] of 16-byte pieces (8 bf16 channels): written once by the producer
//           (k_gln_p / k_featurize_p / a blend epilogue), so the consumer's patch staging is a pure copy - no fp32 loads, no
//           VALU split, no LDS stores from registers: every thread issues `buffer_load_dwordx4 ... lds` (LDS-DMA) with a
//           per-lane source offset; halo columns / rows outside the tensor use an out-of-range offset, which the buffer
//           bounds check turns into zeros.  History rows (t < 0) come from the previous ring slot of the same allocation.
//   output  "R layout"  R[b][octet][pos][8] fp32 (pos = t * oT + oo + m): a lane holds 4 consecutive channels x 4 octets of
//           one position, so the epilogue is four 16-byte stores per tile and a wave store covers 1 KB of contiguous
//           memory (the F-innermost fp32 layout of the first generation wrote 132 / 68-byte rows: 1.6-2.3x write
//           amplification).  Transposed convolutions write their two output-frequency parities into separate halves of a
//           row (parity-planar) instead of interleaving 4-byte elements.
//           Alternatives: P layout directly (activation only, or the fused decoder skip gate), statistics only.
//   tiling  up to 12 column tiles per wave (512 VGPRs at one workgroup per CU): one workgroup can own a whole stream, so the
//           time-tap halo is staged once per stream, a weight fragment feeds up to 72 MFMAs, and B = 256 streams are exactly
//           one round of 256 CUs.
// Reference semantics: TemporalConv2d / TemporalConvTranspose2d / the 1x1 skip convolutions, CRN.py:290-401.
#pragma once
#include <hip/hip_runtime.h>

#include "conv_p_args.h"
#include "split_bf16.h"

namespace se {



__device__ __forceinline__ float convp_act(float v, int act) {
    return act == 1 ? fmaxf(v, 0.0f) : (act == 2 ? (v > 0.0f ? v : expf(v) - 1.0f) : v);
}

// fp32 x 8 -> PL planes of 8 bf16 (or one fp16 plane)
template <int PL>
__device__ __forceinline__ void split_store8(const float (&v)[8], uint4 *dst, long plane_stride) {
    if (PL == 1) {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        h8 hv;
#pragma unroll
        for (int c = 0; c < 8; c++) hv[c] = (_Float16)v[c];
        dst[0] = __builtin_bit_cast(uint4, hv);
    } else {
        bf16x8 h, m, l;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            __bf16 hh, mm, ll;
            split3(v[c], hh, mm, ll);
            h[c] = hh; m[c] = mm; l[c] = ll;
        }
        dst[0] = __builtin_bit_cast(uint4, h);
        dst[plane_stride] = __builtin_bit_cast(uint4, m);
        if (PL == 3) dst[2 * plane_stride] = __builtin_bit_cast(uint4, l);
    }
}

__device__ __forceinline__ void convp_stats_store(float *stats, int nslot, int slot0, float s, float q, float *red /*[8]*/, int b) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); q += __shfl_down(q, off, 64); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { red[wave * 2] = s; red[wave * 2 + 1] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float *o = stats + ((long)b * nslot + slot0 + blockIdx.x) * 2;
        o[0] = (red[0] + red[2]) + (red[4] + red[6]);
        o[1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
}

// -DSE_CP_TRACE (diagnostic build of the instances of interest + se_engine.hip): thread 0 of every workgroup sums its s_memtime deltas per
// phase and adds them to a.trace; printed per launch site at process exit.  Read the shares, not the run time of that build.
#ifdef SE_CP_TRACE
#define CPT(i) do { unsigned long long _n; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_n) :: "memory"); \
                    __builtin_amdgcn_sched_barrier(0); tr[i] += _n - tlast; tlast = _n; } while (0)
#else
#define CPT(i)
#endif

template <int NTAP, int NT, int CO, int PL>
__global__ __launch_bounds__(256, NT <= 6 ? 2 : 1) void k_conv_p(ConvPArgs a) {
#ifdef SE_CP_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast) :: "memory");
#endif
    // LDS patch [PL][CO][Npos] of 16-byte pieces (plane-major: consecutive LDS-DMA lanes copy consecutive 16-byte pieces of one
    // plane row, i.e. whole cache lines; an interleaved [pos][plane] image made every wave instruction touch three scattered
    // runs and staged at 9 GB/s per CU)
    extern __shared__ __align__(16) uint4 planes[];
    constexpr int NPAIR = (NTAP * CO + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;  // provably wave-uniform (LDS-DMA destination, tile ownership)
    const int b = blockIdx.y;
    float bmy = 0.f, biy = 1.f, bmu = 0.f, biu = 1.f;
    if (a.out_mode == kPOutBlend) {  // this stream's norm statistics for the fused skip gate (LDS is still free)
        float *sm = reinterpret_cast<float *>(planes);
        slab_mean_inv(a.bl_sy, b, sm, bmy, biy);
        slab_mean_inv(a.bl_su, b, sm + 2, bmu, biu);
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
    const float invFP = 1.0f / (float)a.FP;

    int lane_base[NT];
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const int pc = min(p0 + (cg + i * NCG) * 32 + l31, p1 - 1);
        const int t = (int)(((float)pc + 0.5f) * invFP), m = pc - t * a.FP;
        lane_base[i] = ((t - ta) * St + (a.deint ? m : a.s * m)) * 16;  // byte offset inside plane 0
    }
    int toffL[NPAIR];  // per lane half: LDS offset of entry 2*step + half = (tap, octet), tap-major
#pragma unroll
    for (int pr = 0; pr < NPAIR; pr++) {
        const int en = min(2 * pr + half, NTAP * CO - 1);  // a padded (zero-weight) entry reads any valid position
        const int tp = en / CO, oc = en - tp * CO;
        toffL[pr] = (oc * Npos + a.rowgrp[tp] * (a.grouped ? RT : a.dil) * St + a.coloff[tp]) * 16;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;

    // ---- LDS-DMA plan: item it = tid + 256 k = (plane, octet-in-chunk, patch position), LDS slot = it ----
    const int items = PL * CO * Npos;
    const int NI = (items + 255) >> 8;
    unsigned voff[kPNiMax];  // byte offset of chunk 0's source piece, or kPOob for the zero halo
    {
        // item it = tid + 256 k = (q * R + r) * St + slot: the patch row index rr = q * R + r and the slot advance by (256 / St, 256 % St)
        // per k with one carry, q = rr / R is the only division left per item, and every offset is 32-bit (24-bit multiplies).
        // (The prologue up to the first staging costs 8-15 k cycles per workgroup, 12 % of all conv_p cycles, and it is NOT this
        // arithmetic: the first version - three reciprocal-multiply divisions and 64-bit indices per item - and this one trace the same
        // (profiles/r03_convp_phase_trace_{serial,incremental_plan}.txt); a host-built offset table traced 1.5-2x worse, its 72 KB of loads
        // per workgroup arriving when every workgroup of the launch starts at once (..._after.txt).)
        const int NGp = a.grouped ? a.ngroup : 1, RTp = a.grouped ? RT : R;
        const float invR = 1.0f / (float)R, invSt = 1.0f / (float)St, invRTp = 1.0f / (float)RTp;
        const long sb = (long)b * a.C8 * PL * a.T * a.Fi;
        const unsigned base_cur = (unsigned)((a.cur_off + sb) * 16), base_prev = (unsigned)((a.prev_off + sb) * 16);
        const bool has_prev = a.prev_off >= 0;
        const int d_r = (int)((256.0f + 0.5f) * invSt), d_slot = 256 - d_r * St;  // uniform
        int rr = (int)(((float)tid + 0.5f) * invSt), slot = tid - rr * St;
        const int rows = PL * CO * R;
#pragma unroll
        for (int k = 0; k < kPNiMax; k++) {
            voff[k] = kPOob;
            if (k < NI) {
                if (rr < rows) {  // it < items
                    const int q = (int)(((float)rr + 0.5f) * invR), r = rr - q * R;  // q = pl * CO + oc
                    const int pl = q / CO, oc = q - pl * CO;
                    const int col = a.deint ? (slot < a.Sh ? 2 * slot : 2 * (slot - a.Sh) + 1) : slot;  // even columns first, then the odd ones
                    const int g = NGp > 1 ? (int)(((float)r + 0.5f) * invRTp) : 0, j = r - g * RTp;
                    const int ts = ta + a.tlo_off + g * a.dil + j;
                    const int fi = col - a.colpad;
                    const bool hist = ts < 0;
                    const bool ok = fi >= 0 && fi < a.Fi && (hist ? (has_prev && ts + a.T >= 0) : ts < a.T);
                    if (ok) {
                        const unsigned row = __umul24((unsigned)(oc * PL + pl), (unsigned)a.T) + (unsigned)(hist ? ts + a.T : ts);
                        voff[k] = (hist ? base_prev : base_cur) + ((__umul24(row, (unsigned)a.Fi) + (unsigned)fi) << 4);
                    }
                }
                slot += d_slot; rr += d_r;
                if (slot >= St) { slot -= St; rr++; }
            }
        }
    }
    const unsigned chunk_bytes = (unsigned)((long)CO * PL * a.T * a.Fi * 16);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(a.xbase), 0, a.xbytes, 0x00020000);
    auto stage = [&](int ch) {
        const unsigned add = (unsigned)ch * chunk_bytes;
#pragma unroll
        for (int k = 0; k < kPNiMax; k++) {
            if (k < NI) {
                const unsigned v = voff[k] == kPOob ? kPOob : voff[k] + add;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(planes + k * 256 + wave * 64), 16, v, 0, 0, 0);
            }
        }
    };

    const uint4 *wxw = a.wx + (long)mt * 64 + l31 * 2 + half;  // + (((ch*NPAIR + pr)*PL + plane)*MT) * 64
    const long wx_plane = (long)MT * 64, wx_pair = PL * wx_plane, wx_chunk = NPAIR * wx_pair;

    const char *ldsb = reinterpret_cast<const char *>(planes);
    const int plane_bytes = CO * Npos * 16;
    // One B fragment = the PL planes of 8 channels at one patch position.  The address is made opaque right before its reads:
    // otherwise the compiler hoists all NPAIR x NT x PL fragment addresses out of the chunk loop (hundreds of registers parked
    // in AGPRs and read back one by one) - one v_add per fragment is free next to six MFMAs.
    auto read_b = [&](int i, int pr, uint4 (&bf)[PL]) {
        int off = lane_base[i] + toffL[pr];
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int p = 0; p < PL; p++) bf[p] = *reinterpret_cast<const uint4 *>(ldsb + off + p * plane_bytes);
    };
    CPT(0);  // prologue: tile geometry + LDS-DMA plan
    for (int ch = 0; ch < a.nchunk; ch++) {
        const uint4 *wc = wxw + ch * wx_chunk;
        uint4 fa_n[PL];
#pragma unroll
        for (int p = 0; p < PL; p++) fa_n[p] = wc[p * wx_plane];  // first weight fragments of the chunk: in flight during staging
        __syncthreads();  // previous chunk fully consumed
        CPT(1);
        stage(ch);
        CPT(2);
        __syncthreads();  // (the compiler drains vmcnt before the barrier: every wave's DMA pieces have landed)
        CPT(3);
        uint4 bcur[PL], bnxt[PL];
        read_b(0, 0, bcur);
#pragma unroll
        for (int pr = 0; pr < NPAIR; pr++) {
            uint4 fa[PL];
#pragma unroll
            for (int p = 0; p < PL; p++) fa[p] = fa_n[p];
            if (pr + 1 < NPAIR) {
#pragma unroll
                for (int p = 0; p < PL; p++) fa_n[p] = wc[(pr + 1) * wx_pair + p * wx_plane];
            }
#pragma unroll
            for (int i = 0; i < NT; i++) {
                // software pipeline, one fragment deep: the LDS reads of the NEXT (tile, K step) are issued before this
                // fragment's MFMAs, whose 100-200 matrix-pipe cycles cover the LDS latency (one wave per SIMD: nothing else would)
                if (i + 1 < NT) read_b(i + 1, pr, bnxt);
                else if (pr + 1 < NPAIR) read_b(0, pr + 1, bnxt);
                __builtin_amdgcn_sched_barrier(0);
                f32x16 c = acc[i];
                if (PL >= 2) {
                    const bf16x8 a0 = __builtin_bit_cast(bf16x8, fa[0]), a1 = __builtin_bit_cast(bf16x8, fa[PL > 1 ? 1 : 0]),
                                 a2 = __builtin_bit_cast(bf16x8, fa[PL > 2 ? 2 : 0]);
                    const bf16x8 b0 = __builtin_bit_cast(bf16x8, bcur[0]), b1 = __builtin_bit_cast(bf16x8, bcur[PL > 1 ? 1 : 0]);
                    if (PL == 3) {
                        const bf16x8 b2 = __builtin_bit_cast(bf16x8, bcur[PL > 2 ? 2 : 0]);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c, 0, 0, 0);  // mid*mid
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, c, 0, 0, 0);  // hi*lo
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, c, 0, 0, 0);  // lo*hi
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c, 0, 0, 0);  // hi*mid
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c, 0, 0, 0);  // mid*hi
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c, 0, 0, 0);  // hi*hi
                } else {
                    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, fa[0]), __builtin_bit_cast(h8, bcur[0]), c, 0, 0, 0);
                }
                acc[i] = c;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < PL; p++) bcur[p] = bnxt[p];
            }
        }
        CPT(4);  // MFMAs issued (the tail retires inside the next barrier segment)
    }
    __syncthreads();  // LDS is free again (statistics scratch)
    CPT(1);
#ifdef SE_CP_TRACE
    struct TraceOut {  // every epilogue returns: the destructor closes the last segment
        unsigned long long *tr, &tlast, *out; int tid;
        __device__ ~TraceOut() {
            unsigned long long _n;
            asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_n) :: "memory");
            tr[5] += _n - tlast;
            if (out && tid == 0) { for (int i = 0; i < 6; i++) atomicAdd(out + i, tr[i]); atomicAdd(out + 7, 1ull); }
        }
    } trace_out{tr, tlast, a.trace, tid};
#endif

    // ---- epilogue: lane = position (column of the 32x32 tiles), registers = GEMM rows (r & 3) + 8 (r >> 2) + 4 half ----
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; r++) bv[r] = a.bias[mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];  // bias is padded to CoPad
#pragma unroll
    for (int r = 0; r < 16; r++) asm volatile("" : "+v"(bv[r]));  // all bias loads ahead of the first store
    float ssum = 0.0f, ssq = 0.0f;
    if (a.out_mode == kPOutBlend) {
        // rows are permuted: register pair (2q, 2q+1) of lane half h = (residualmask, residual) of channel mt*16 + h*8 + q
        const int c0 = mt * 16 + half * 8;
        float cnw[8], cnb[8], cmw[8], cmb[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int c = min(c0 + q, a.Cy - 1);
            cnw[q] = a.bl_nw[c]; cnb[q] = a.bl_nb[c]; cmw[q] = a.bl_mnw[c]; cmb[q] = a.bl_mnb[c];
        }
        const float *ydb = a.bl_ydec + (long)b * a.bl_stream + (long)(c0 >> 3) * a.T * a.bl_oT * 8;
        uint4 *ypb = a.yp + (long)b * a.yp_stream + (long)(c0 >> 3) * PL * a.T * a.Fy;
        const long plane_stride = (long)a.T * a.Fy;
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int p = p0 + (cg + i * NCG) * 32 + l31;
            if (p >= p1 || c0 >= a.Cy) continue;
            const int t = (int)(((float)p + 0.5f) * invFP), m = p - t * a.FP;
            const int f = a.oo + m;
            float yv[8];
            if (f < a.bl_Fo) {  // the transposed convolution gives 2 Fi - 1 bins: zero beyond (CRN.py:389-392)
                const float4 *src = reinterpret_cast<const float4 *>(ydb + ((long)t * a.bl_oT + (f & 1) * a.bl_Fh + (f >> 1)) * 8);
                const float4 u0 = src[0], u1 = src[1];
                yv[0] = u0.x; yv[1] = u0.y; yv[2] = u0.z; yv[3] = u0.w; yv[4] = u1.x; yv[5] = u1.y; yv[6] = u1.z; yv[7] = u1.w;
            }
            float o[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const float u = acc[i][2 * q] + bv[2 * q];
                const float v = convp_act(acc[i][2 * q + 1] + bv[2 * q + 1], a.act);
                const float yn = f < a.bl_Fo ? (yv[q] - bmy) * biy * cnw[q] + cnb[q] : 0.0f;
                const float un = (u - bmu) * biu * cmw[q] + cmb[q];
                const float g = 1.0f / (1.0f + expf(-un));
                o[q] = (c0 + q < a.Cy) ? g * v + (1.0f - g) * yn : 0.0f;
            }
            split_store8<PL>(o, ypb + (long)t * a.Fy + f, plane_stride);
        }
        return;
    }
    if (a.out_mode == kPOutP) {
        // rows are permuted: registers 0..7 / 8..15 of lane half h = the 8 channels of octet mt*4 + 2h / mt*4 + 2h + 1
        const long plane_stride = (long)a.T * a.Fy;
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int p = p0 + (cg + i * NCG) * 32 + l31;
            if (p >= p1) continue;
            const int t = (int)(((float)p + 0.5f) * invFP), m = p - t * a.FP;
            if (m >= a.valid_m) continue;
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const int oct = mt * 4 + 2 * half + hh;
                if (oct * 8 >= a.Cy) continue;
                float o[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int row = mt * 32 + half * 16 + hh * 8 + q;  // logical channel of this register
                    float v = acc[i][hh * 8 + q] + bv[hh * 8 + q];
                    if (row >= a.relu_lo && row < a.relu_hi) v = convp_act(v, a.act);
                    o[q] = row < a.Cy ? v : 0.0f;
                }
                split_store8<PL>(o, a.yp + (long)b * a.yp_stream + (long)oct * PL * plane_stride + (long)t * a.Fy + a.oo + m, plane_stride);
            }
        }
        return;
    }
    if (a.out_mode == kPOutPre) {
        // <= 8 channels: GEMM rows 0-3 / 8-11 are channels 0-7, i.e. registers 0-7 of lane half 0 (half 1 holds padding rows)
        float *yb = a.y + (long)b * a.y_stream;
        const int C = a.Cy;
        const float *gw = a.gatew;
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int p = p0 + (cg + i * NCG) * 32 + l31;
            if (p >= p1 || half) continue;
            const int t = (int)(((float)p + 0.5f) * invFP), m = p - t * a.FP;
            float x[8], o[8];
#pragma unroll
            for (int c = 0; c < 8; c++) x[c] = c < C ? convp_act(acc[i][c] + bv[c], a.act) : 0.0f;
#pragma unroll
            for (int c = 0; c < 8; c++) {
                float tr = 0.0f, gt = 0.0f;
                if (c < C) {
                    tr = gw[2 * C * C + c];
                    gt = gw[2 * C * C + C + c];
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        if (j < C) { tr += gw[c * C + j] * x[j]; gt += gw[C * C + c * C + j] * x[j]; }
                }
                const float v = c < C ? tr * (1.0f / (1.0f + expf(-gt))) : 0.0f;
                o[c] = v;
                ssum += v; ssq += v * v;
            }
            float4 *dst = reinterpret_cast<float4 *>(yb + ((long)t * a.oT + a.oo + m) * 8);
            dst[0] = make_float4(o[0], o[1], o[2], o[3]);
            dst[1] = make_float4(o[4], o[5], o[6], o[7]);
        }
        if (a.stats) convp_stats_store(a.stats, a.stats_nslot, a.stats_slot0, ssum, ssq, reinterpret_cast<float *>(planes), b);
        return;
    }
    if (a.out_mode == kPOutGate) {
        // rows are permuted like the blend epilogue: register pair (2q, 2q+1) of lane half h = (conv_trans, conv_gated) of
        // channel gate_c0 + mt*16 + h*8 + q; the product goes to the R layout with its statistics for the block's gLN
        const int c0 = mt * 16 + half * 8;
        float *yb = a.y + (long)b * a.y_stream + (long)((a.gate_c0 + c0) >> 3) * a.y_npos * 8;
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int p = p0 + (cg + i * NCG) * 32 + l31;
            if (p >= p1 || c0 >= a.Cy) continue;
            const int t = (int)(((float)p + 0.5f) * invFP), m = p - t * a.FP;
            float o[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const float u = acc[i][2 * q] + bv[2 * q];
                const float v = acc[i][2 * q + 1] + bv[2 * q + 1];
                const float x = (c0 + q < a.Cy) ? u * (1.0f / (1.0f + expf(-v))) : 0.0f;
                o[q] = x;
                ssum += x; ssq += x * x;
            }
            float4 *dst = reinterpret_cast<float4 *>(yb + ((long)t * a.oT + a.oo + m) * 8);
            dst[0] = make_float4(o[0], o[1], o[2], o[3]);
            dst[1] = make_float4(o[4], o[5], o[6], o[7]);
        }
        if (a.stats) convp_stats_store(a.stats, a.stats_nslot, a.stats_slot0, ssum, ssq, reinterpret_cast<float *>(planes), b);
        return;
    }
    // R layout (or statistics only): register group q = r >> 2 holds rows 8 q + 4 half + {0..3} of M tile mt
    float *yb = a.y ? a.y + (long)b * a.y_stream : nullptr;
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const int p = p0 + (cg + i * NCG) * 32 + l31;
        if (p >= p1) continue;
        const int t = (int)(((float)p + 0.5f) * invFP), m = p - t * a.FP;
        if (m >= a.valid_m) continue;
        const long opos = (long)t * a.oT + a.oo + m;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int row0 = mt * 32 + 8 * q + 4 * half;
            if (row0 >= a.Co) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int row = row0 + e;
                float x = acc[i][4 * q + e] + bv[4 * q + e];
                if (row >= a.relu_lo && row < a.relu_hi) x = convp_act(x, a.act);
                x = row < a.Co ? x : 0.0f;
                v[e] = x;
                if (row >= a.stats_lo && row < a.stats_hi && !(a.par_rows && (row & 1) && m >= a.FP - 1)) { ssum += x; ssq += x * x; }
            }
            if (yb) *reinterpret_cast<float4 *>(yb + ((long)(row0 >> 3) * a.y_npos + opos) * 8 + (row0 & 7)) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    if (a.stats) convp_stats_store(a.stats, a.stats_nslot, a.stats_slot0, ssum, ssq, reinterpret_cast<float *>(planes), b);
}

// ---------------------------------------------------------------------------------------------------------------------
// Producers / consumers of the two layouts
// ---------------------------------------------------------------------------------------------------------------------
// features (CRN.py:463-467) straight into the P layout of the first encoder block: one octet = [mag_0..M-1, dphi_1..M-1, 0...]

template <int PL>
__global__ __launch_bounds__(256) void k_featurize_p(FeatPArgs a) {
    const int b = blockIdx.y, TF = a.T * a.F;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= TF) return;
    const int t = i / a.F, f = i - t * a.F;
    const cf2 *s = a.spec + (long)b * a.sB + (long)t * a.sT + (long)f * a.sF;
    float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float ang0 = 0.0f;
    for (int m = 0; m < a.M; m++) {
        const cf2 v = s[(long)m * a.sM];
        o[m] = sqrtf(v.x * v.x + v.y * v.y + 1e-10f);
        const float ang = a.atan2_phase ? atan2f(v.y, v.x) : atanf(v.y / (v.x + kEps) + kEps);
        if (m == 0) ang0 = ang;
        else o[a.M + m - 1] = ang0 - ang;
    }
    split_store8<PL>(o, a.out + (long)b * a.out_stream + i, TF);
}

// fp32 [b][C][T*F] with C <= 8 -> octet 0 of the P layout (CRN_ELU / student: the fp32 pre-conv chain hands over here)
template <int PL>
__global__ __launch_bounds__(256) void k_f32_to_p(F32ToPArgs a) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.TF) return;
    const float *x = a.x + (long)b * a.C * a.TF + i;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 8; c++)
        if (c < a.C) v[c] = x[(long)c * a.TF];
    split_store8<PL>(v, a.out + (long)b * a.out_stream + i, a.TF);
}

// gLN from the producing convolution's partial statistics: R layout -> P layout (normalise, per-channel affine, split)

template <int PL>
__global__ __launch_bounds__(256) void k_gln_p(GlnPArgs a) {
    __shared__ float sm[2];
    const int b = blockIdx.z, o = blockIdx.y;
    float mean, inv;
    slab_mean_inv(a.st, b, sm, mean, inv);
    const int TF = a.T * a.F;
    float sc[8], sh[8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int ch = o * 8 + c;
        const float w = ch < a.C ? a.w[ch] : 0.0f, bb = ch < a.C ? a.b[ch] : 0.0f;
        sc[c] = inv * w;
        sh[c] = bb - mean * inv * w;
    }
    const float invF = 1.0f / (float)a.F;
    const float *xb = a.x + (long)b * a.x_stream + (long)o * a.T * a.in_oT * 8;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < TF; i += gridDim.x * 256) {
        const int t = (int)(((float)i + 0.5f) * invF), f = i - t * a.F;
        const float4 *src = reinterpret_cast<const float4 *>(xb + ((long)t * a.in_oT + f) * 8);
        const float4 u0 = src[0], u1 = src[1];
        float v[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
#pragma unroll
        for (int c = 0; c < 8; c++) v[c] = v[c] * sc[c] + sh[c];  // (x - mean) * inv * w + b; zero for padded channels
        if (a.res) {  // + x: the planes of x, summed from the smallest, give x back exactly
            const uint4 *rp = a.res + (long)b * a.res_stream + (long)o * PL * TF + i;
            float r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int pl = PL - 1; pl >= 0; pl--) {
                const uint4 u = rp[(long)pl * TF];
                if (PL == 1) {
                    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                    const h8 hv = __builtin_bit_cast(h8, u);
#pragma unroll
                    for (int c = 0; c < 8; c++) r[c] += (float)hv[c];
                } else {
                    const bf16x8 bv8 = __builtin_bit_cast(bf16x8, u);
#pragma unroll
                    for (int c = 0; c < 8; c++) r[c] += (float)bv8[c];
                }
            }
#pragma unroll
            for (int c = 0; c < 8; c++) v[c] += r[c];
        }
        if (a.mode == 0) split_store8<PL>(v, a.y + (long)b * a.y_stream + (long)o * PL * TF + i, TF);
        else split_store8<PL>(v, a.y + ((long)(b * a.T + t) * a.C8 + o) * a.F + f, a.y_plane);
    }
}

// exact two-pass gLN(last=True) of the fc output (CRN.py:127-129, 279): x [b][T][D] fp32, d = c * F + f (the reference's
// feature order) -> decoder input in the P layout

template <int PL>
__global__ __launch_bounds__(1024) void k_gln2_p(Gln2PArgs a) {
    __shared__ double red[16];
    const int b = blockIdx.x;
    const int D = a.C * a.F;
    const long n = (long)a.T * D;
    const float *x = a.x + (long)b * n;
    float mean, inv;
    stream_stats(x, n, red, mean, inv, a.eps_mode);
    const int TF = a.T * a.F, OF = a.C8 * a.F;
    for (int i = threadIdx.x; i < a.T * OF; i += 1024) {  // i = (t * C8 + o) * F + f: consecutive lanes = consecutive f
        const int t = i / OF, of = i - t * OF, o = of / a.F, f = of - o * a.F;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int ch = o * 8 + c;
            const int d = min(ch, a.C - 1) * a.F + f;
            v[c] = ch < a.C ? (x[(long)t * D + d] - mean) * inv * a.w[d] + a.b[d] : 0.0f;
        }
        split_store8<PL>(v, a.y + (long)b * a.y_stream + (long)o * PL * TF + (long)t * a.F + f, TF);
    }
}

// last gLN + decompress_cIRM + complex multiply (CRN.py:491-495) from the R layout of the final transposed convolution,
// whose GEMM rows (2c + parity) are the two output-frequency parities of mask channel c: slots [c0 even, c0 odd, c1 even, c1 odd]

template <int DUMMY>
__global__ __launch_bounds__(256) void k_final_mask_p(MaskPArgs a) {
    __shared__ float sm[2];
    const int b = blockIdx.y;
    float mean, inv;
    slab_mean_inv(a.st, b, sm, mean, inv);
    const int TF = a.T * a.F;
    const float *y = a.y + (long)b * a.y_stream;
    const float w0 = a.nw[0], w1 = a.nw[1], b0 = a.nb[0], b1 = a.nb[1];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < TF; i += gridDim.x * 256) {
        const int t = i / a.F, f = i - t * a.F;
        const float *p = y + ((long)t * a.Fh + (f >> 1)) * 8 + (f & 1);
        const float mr = decompress_cirm((p[0] - mean) * inv * w0 + b0);
        const float mi = decompress_cirm((p[2] - mean) * inv * w1 + b1);
        const cf2 n = a.spec[(long)b * a.sB + (long)t * a.sT + (long)f * a.sF];
        a.out[(long)b * a.oB + (long)t * a.oT + (long)f * a.oF] = cf2{mr * n.x - mi * n.y, mi * n.x + mr * n.y};
    }
}

}  // namespace se
