// se_aux.hip - the kernels of the path that do COMPLEX arithmetic (LDS FFTs of the STFT / iSTFT, mask application),
// compiled as their own translation unit with -fno-slp-vectorize.
//
// Why: with SLP vectorisation hipcc turns the (re, im) float pairs of these kernels into packed-FP32 VALU instructions
// with operand swizzles (v_pk_mul_f32 / v_pk_fma_f32 ... op_sel:[1,1], neg_lo/neg_hi, v_pk_mov_b32).  On MI355X those
// kernels returned wrong values - transiently, in 35 % of the launches - whenever their waves shared a SIMD with waves of
// this engine's MFMA kernels launched from another stream (signal / window / twiddle tables in LDS intact,
// a redundant re-execution of one radix-4 pass inside the kernel differs, scalar-FMA and integer LDS victims never fail,
// plain op_sel_hi-only packed ops as in k_conv_small never failed either).  Built without SLP the same kernels contain
// no packed-FP32 instruction, cost the same time (STFT 56 vs 57 us) and are exact under co-execution
// (tests/manual_coexec_fft.py: 0 of 800 launches differ).
#include <hip/hip_runtime.h>
#define SE_AUX_KERNELS 1
#include "fft_lds.h"
#include "norm.hip.h"
#include "stft.hip.h"
#include "fsn_mask.hip.h"

namespace se {

void launch_k_stft(dim3 grid, size_t lds, hipStream_t st, const StftArgs &a) { hipLaunchKernelGGL(k_stft, grid, dim3(256), lds, st, a); }

void launch_k_istft(dim3 grid, size_t lds, hipStream_t st, const IstftArgs &a) { hipLaunchKernelGGL(k_istft, grid, dim3(256), lds, st, a); }

void launch_k_overlap_avg(dim3 grid, hipStream_t st, const float *yseg, float *out, int Nseg, int K, long L, long skip, const long *Lrow) {
    hipLaunchKernelGGL(k_overlap_avg, grid, dim3(256), 0, st, yseg, out, Nseg, K, L, skip, Lrow);
}

void launch_k_final_mask_ew(dim3 grid, hipStream_t st, const MaskEwArgs &a) { hipLaunchKernelGGL(k_final_mask_ew, grid, dim3(256), 0, st, a); }

void launch_k_fsn_mask(dim3 grid, hipStream_t st, const FsnMaskArgs &a) { hipLaunchKernelGGL(k_fsn_mask, grid, dim3(256), 0, st, a); }

void aux_set_fft_lds(int stft_bytes, int istft_bytes) {
    // engines / signal handles of different STFT sizes share the kernels: the opt-in only ever grows
    // (a geometry whose segment does not fit the 160 KB of LDS - FullSubNet's train=True engine with one N*T-frame "segment" -
    // never launches these kernels and must not poison the opt-in of the others)
    static int cur_stft = 0, cur_istft = 0;
    const int kMaxLds = 160 * 1024;
    if (stft_bytes <= kMaxLds && stft_bytes > cur_stft) cur_stft = stft_bytes;
    if (istft_bytes <= kMaxLds && istft_bytes > cur_istft) cur_istft = istft_bytes;
    stft_bytes = cur_stft; istft_bytes = cur_istft;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_stft), hipFuncAttributeMaxDynamicSharedMemorySize, stft_bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_istft), hipFuncAttributeMaxDynamicSharedMemorySize, istft_bytes);
}

}  // namespace se
