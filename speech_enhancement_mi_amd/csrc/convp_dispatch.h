// convp_dispatch.h - host entry points of the plane-layout convolution path (conv_p.hip.h), compiled in se_convp*.o.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_p_args.h"

namespace se {

// k_conv_p<NTAP, NT, CO, PL>: 0 = launched, 1 = no such instance
int conv_p_launch(int ntap, int NT, int CO, int PL, dim3 grid, size_t lds, hipStream_t st, const ConvPArgs &a);
bool conv_p_has_instance(int ntap, int NT, int CO);
void conv_p_set_attributes();
void launch_k_featurize_p(int PL, dim3 grid, hipStream_t st, const FeatPArgs &a);
void launch_k_gln_p(int PL, dim3 grid, hipStream_t st, const GlnPArgs &a);
void launch_k_f32_to_p(int PL, dim3 grid, hipStream_t st, const F32ToPArgs &a);
void launch_k_gln2_p(int PL, dim3 grid, hipStream_t st, const Gln2PArgs &a);
void launch_k_final_mask_p(dim3 grid, hipStream_t st, const MaskPArgs &a);
// k_skip_p<PL, KP> (skip_p.hip.h): one workgroup of 512 threads per stream; 0 = launched, 1 = no such instance
int launch_k_skip_p(int PL, int KP, int batch, size_t lds, hipStream_t st, const struct SkipPArgs &a);
void skip_p_set_attributes();

}  // namespace se
