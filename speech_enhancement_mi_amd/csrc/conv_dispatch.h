// conv_dispatch.h - host-side entry points of the convolution kernels, which are compiled in their own translation unit
// (se_conv.hip: 100+ template instances) so that the engine and the kernels build in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_args.h"

namespace se {

// k_conv_x6<NTAP, NT, CO, PL>: returns 0 on launch, 1 when no instance exists for the combination
int conv_x6_launch(int ntap, int NT, int CO, int PL, dim3 grid, size_t lds, hipStream_t st, const ConvX6Args &xa);
// k_conv_igemm<NTAP, NT> (NT >= 1) or k_conv_small<NTAP, CoPad/4> (NT == 0)
int conv_igemm_launch(int ntap, int NT, int CoPad, dim3 grid, size_t lds, hipStream_t st, const ConvArgs &a);
void conv_set_attributes();   // opt in to large dynamic LDS for every instance
void conv_x6_trace_dump();    // -DSE_X6_TRACE builds: prints the per-phase cycle trace

}  // namespace se
