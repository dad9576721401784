// conv_args.h - argument structs and staging budgets of the convolution kernels (conv_igemm.hip.h, conv_x6.hip.h), shared by the
// engine (se_engine.hip: planning, launches through conv_dispatch.h) and the kernel translation unit (se_conv.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "norm.hip.h"

namespace se {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxTaps = 25;  // 5x5 frequency-dilated preconv blocks of CRN_ELU.py:335-340

struct ConvArgs {
    const float *x;      // [B][Ci][T][Fi] current step
    const float *xprev;  // same tensor of the previous step (rows t<0), or nullptr
    const float *w;      // [nchunk][ntap][CC][CoPad] pre-arranged on the host
    const float *bias;   // [Co]
    float *y;            // [B][Co][T][Fy]
    int Ci, Co, CoPad, T, Fi, FP, Fy;
    int s, os, oo;       // input column = s*m + coloff ; output column = os*m + oo
    int colpad;          // patch column c holds input freq c - colpad
    int tlo_off;         // first row group holds source times ta + tlo_off + j
    int ngroup, dil;     // time taps: row group g holds source times ta + tlo_off + g*dil + j, j in [0, tb-ta]
    int grouped;         // 1: patch rows = ngroup * (tb-ta+1) (one block per time tap; wins when dil is large)
                         // 0: patch rows = (tb-ta+1) + (ngroup-1)*dil (one contiguous time range)
    int ntap;
    int rowgrp[kMaxTaps];  // row group (time tap index) of each tap
    int coloff[kMaxTaps];
    int CC, nchunk, tiles_per_wg, St;
    int relu_lo, relu_hi;  // GEMM rows in [relu_lo, relu_hi) get the activation `act`
    int act;               // 1 = ReLU (CRN.py), 2 = ELU (CRN_ELU.py:226,280)
    // gate_pairs: GEMM rows (2c, 2c+1) hold conv_trans_c and conv_gated_c of the gated 1x1 pair
    // (CRN_ELU.py:240: out = conv_trans(out) * sigmoid(conv_gated(out))); the epilogue writes channel c = trans * sigmoid(gated)
    int gate_pairs;
    int Cy, cy0;           // channels of y per stream and channel offset of this launch (row -> channel cy0 + row[/2])
    // par_rows: GEMM rows (2c, 2c+1) are the EVEN and ODD output-frequency parity of channel c of a transposed convolution
    // whose two tap sets were merged into one launch (weights are zero where a parity does not use a tap): row 2c+p is
    // stored at column os*m + oo + p of channel c.  Used for narrow decoder blocks (<= 16 channels), where each parity alone
    // would leave half of the 32-row MFMA tile empty.
    int par_rows;
    // k_conv_small only: fused gated 1x1 pair on the activated outputs (all Co channels of a position live in one
    // thread): gatew = [trans Co x Co | gated Co x Co | trans bias Co | gated bias Co], nullptr = off
    const float *gatew;
    // per-workgroup partial (sum, sum of squares) of the stored activations of channels [stats_lo, stats_hi),
    // written to stats[(b*stats_nslot + stats_slot0 + blockIdx.x)*2 + {0,1}] for the global layer norm that
    // follows every block (CRN.py:135-149); nullptr = off.  One slot per workgroup -> deterministic.
    float *stats;
    int stats_nslot, stats_slot0, stats_lo, stats_hi;
    // Decoder skip gate fused into the 1x1 skip convolution (k_conv_x6 only; CRN.py:387-396).  GEMM rows (2c, 2c+1) hold
    // residualmask_c and residual_c of the skip tensor; the epilogue writes
    //     out_c = m * act(residual_c) + (1 - m) * pad(gLN(ydec_c)),   m = sigmoid(gLN(residualmask_c))
    // where the statistics of residualmask come from a stats-only pass of the same convolution (y == nullptr) and those
    // of ydec (the transposed convolution of this block) from its epilogue partials.  blend == 0: off.
    int blend;
    const float *bl_ydec;                    // [B][Cy][T][bl_Fo]
    const float *bl_nw, *bl_nb, *bl_mnw, *bl_mnb;  // [Cy] norm / residualnorm affine
    SlabStats bl_sy, bl_su;
    int bl_Fo;
};

constexpr int kPatchPerThread = 16;   // patch elements staged per thread per chunk  (chunk patch <= 4096 floats)
constexpr int kWeightPerThread = 8;   // float4 weight slots per thread per chunk      (chunk slab  <= 8192 floats)

constexpr int kX6PosPerThread = 4;  // (patch position, octet) items per thread: CO*R*St <= 1024 (host-checked)

struct ConvX6Args {
    ConvArgs c;        // geometry, x / xprev / bias / y / stats exactly as for k_conv_igemm (c.w unused, c.CC == 8)
    const uint4 *wx;   // [nchunk][nstep][3][MT][64] fragments of 16 B
#ifdef SE_X6_TRACE
    int trace_slot;
#endif
};

}  // namespace se
