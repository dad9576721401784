// conv_p_args.h - argument structs of the plane-layout convolution path (kernels: conv_p.hip.h), shared with the engine.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_args.h"

namespace se {

constexpr int kPNiMax = 20;   // LDS-DMA instructions per thread per chunk: PL * CO * Npos <= 256 * kPNiMax (host-checked)
constexpr unsigned kPOob = 0xFFFFFFF0u;  // buffer offset beyond every ring: the bounds check returns zeros

enum ConvPOut : int { kPOutR = 0, kPOutP = 1, kPOutBlend = 2, kPOutStats = 3, kPOutGate = 4, kPOutPre = 5 };

struct ConvPArgs {
    // ---- input: P layout, one allocation holding every ring slot ----
    const uint4 *xbase;     // ring base
    unsigned xbytes;        // bytes of the ring (buffer descriptor range)
    long cur_off, prev_off; // uint4 offsets of the current / history slot inside the ring (prev_off < 0: no history)
    int C8;                 // channel octets of the input tensor
    // ---- geometry (as ConvArgs) ----
    int Ci, Co, CoPad, T, Fi, FP;
    int s, colpad, tlo_off, ngroup, dil, grouped, ntap;
    int rowgrp[kMaxTaps], coloff[kMaxTaps];
    int nchunk, tiles_per_wg, St;
    unsigned long long *trace;  // -DSE_CP_TRACE builds: 8 cycle sums per launch site (nullptr otherwise; never read by the kernel)
    int deint, Sh;          // deint = 1 (stride-2 convolutions): an LDS patch row holds its even columns first, then its odd ones (Sh = (St+1)/2
                            // even slots): column 2m + coloff of lane m becomes slot m + const, a unit-stride ds_read_b128 walk (coloff[] is
                            // then already the slot offset (c & 1) * Sh + (c >> 1)); the HBM layout is unchanged - only the LDS-DMA plan permutes
    const uint4 *wx;        // [nchunk][npair][PL][MT][64] weight fragments
    const float *bias;      // [CoPad] in GEMM-row order
    int act;                // activation on rows [relu_lo, relu_hi): 1 ReLU, 2 ELU
    int relu_lo, relu_hi;
    // ---- output ----
    int out_mode;           // ConvPOut
    float *y;               // kPOutR: R layout [b][Co8][npos][8]
    long y_stream;          // floats per stream
    int y_npos;             // positions per octet (T * oT)
    int oT, oo;             // output position = t * oT + oo + m
    int row_perm;           // 1: GEMM rows are permuted so that a lane holds 8 consecutive channels (P-layout outputs)
    uint4 *yp;              // kPOutP / kPOutBlend: P layout [b][Co8][PL][T][Fy]
    long yp_stream;         // uint4 per stream
    int Fy;                 // row length of the P output; position (t, m) -> column oo + m
    int Cy;                 // channels of the output tensor
    int gate_c0;            // kPOutGate: rows (2c, 2c+1) = (conv_trans, conv_gated) of channel gate_c0 + c (CRN_ELU.py:223-224,240);
                            // y[gate_c0 + c] = trans * sigmoid(gated) in the R layout, statistics of the stored values
    // statistics of the stored values (rows [stats_lo, stats_hi)) -> stats[(b*nslot + slot0 + blockIdx.x)*2 + {0,1}]
    float *stats;
    int stats_nslot, stats_slot0, stats_lo, stats_hi;
    int valid_m;            // positions with m >= valid_m are not stored / counted (odd parity of a transposed conv: Fi - 1)
    int par_rows;           // 1: GEMM rows (2c, 2c+1) = even / odd output-frequency parity of channel c of a transposed convolution
                            // merged into one launch (zero weights where a parity does not use a tap); the odd row of the last
                            // position (m == FP - 1) does not exist and is left out of the statistics
    const float *gatew;     // kPOutPre (5x5 pre-conv blocks of CRN_ELU, <= 8 channels): [trans Cy x Cy][gated Cy x Cy][trans bias][gated bias];
                            // GEMM rows 0-3 / 8-11 = channels 0-7, so lane half 0 holds every channel of its position in registers 0-7 and
                            // the gated 1x1 pair is channel mixing in registers: y = trans(act(conv)) * sigmoid(gated(act(conv))) -> R layout
    // ---- fused decoder skip gate (kPOutBlend): rows (2c, 2c+1) = (residualmask_c, residual_c) ----
    const float *bl_ydec;   // R layout of the transposed convolution of this block: [b][Cy8][T * bl_oT][8]
    long bl_stream;
    int bl_oT, bl_Fh, bl_Fo; // decoder output column f lives at t * bl_oT + (f & 1) * bl_Fh + (f >> 1); f >= bl_Fo -> zero pad
    const float *bl_nw, *bl_nb, *bl_mnw, *bl_mnb;
    SlabStats bl_sy, bl_su;
};

struct FeatPArgs {
    const cf2 *spec;
    long sB, sM, sT, sF;
    uint4 *out;      // P[b][0][pl][t][f]
    long out_stream; // uint4 per stream
    int M, T, F, atan2_phase;
};

struct F32ToPArgs {   // fp32 [b][C][T*F] (C <= 8) -> P layout octet 0: the hand-over from the fp32 pre-conv chain of CRN_ELU
    const float *x;
    uint4 *out;
    long out_stream;  // uint4 per stream
    int C, TF;
};

struct GlnPArgs {
    const float *x;     // R layout [b][C8][npos_in][8]
    long x_stream;      // floats per stream
    int C, C8, T, F;    // output geometry: positions (t, f), t < T, f < F
    int in_oT;          // input position of (t, f) = t * in_oT + f
    const float *w, *b; // [C]
    SlabStats st;
    const uint4 *res;   // optional residual in the P layout of the output (CRN_ELU pre-conv blocks: x = m(x) + x), added after the norm
    long res_stream;    // uint4 per stream
    uint4 *y;           // mode 0: P[b][o][pl][t][f]   mode 1: A planes of the GRU input GEMM, [pl][b*T + t][o][f] (K = C8*F*8)
    long y_stream;      // mode 0: uint4 per stream
    long y_plane;       // mode 1: uint4 per plane (= B*T*C8*F)
    int mode;
};

struct Gln2PArgs {
    const float *x;
    const float *w, *b;  // [D]
    uint4 *y;            // P[b][o][pl][t][f]
    long y_stream;
    int T, F, C, C8, eps_mode;
};

struct MaskPArgs {
    const float *y;   // R layout [b][1][T * Fh][8]
    long y_stream;
    int Fh;
    const float *nw, *nb;
    SlabStats st;
    const cf2 *spec;
    long sB, sT, sF;
    cf2 *out;
    long oB, oT, oF;
    int T, F;
};

struct SkipPArgs {
    const uint4 *x;       // res: P layout [b][C8][PL][T*F] of this ring slot
    long x_stream;        // uint4 per stream
    int C, C8, TF;        // channels (Cin = Cout = C), octets, positions per stream (T * Fr)
    int T, Fr;
    const uint4 *wx;      // [MT2][KP][PL][64] fragments: M tiles [0, MTh) = residualmask rows, [MTh, 2 MTh) = residual rows,
                          // rows permuted so that register r of lane half h is channel mt*32 + 16 h + r
    const float *cst;     // [6][Cp] per channel (Cp = MTh*32): residualmask bias, residual bias, norm w, norm b, residualnorm w, b
    int MTh, KP;
    int act;
    const float *ydec;    // R layout of this block's transposed convolution [b][C8][T * 2 Fh][8], parity-planar
    long y_stream;
    int Fh, Fo;           // decoder column f at t * 2 Fh + (f & 1) * Fh + (f >> 1); f >= Fo: zero pad (CRN.py:389-392)
    SlabStats sy;         // statistics of ydec from its producers
    int eps_mode;
    uint4 *out;           // P layout [b][C8][PL][T*Fr]
    long out_stream;
};

}  // namespace se
