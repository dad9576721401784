// fsn_mask.hip.h - FullSubNet mask application (complex arithmetic: kernel body is compiled in se_aux.hip, see there).
#pragma once
#include <hip/hip_runtime.h>
#include "fft_lds.h"
#include "norm.hip.h"

namespace se {

// ---- cIRM decompress + complex multiply with mic 0 (fullsubnet.py:949-953) ---------------------------------------------------
struct FsnMaskArgs {
    const float *mask;  // [B*F][2][T]
    const float *re, *im;  // mic-0 spectrum (b, t, f) at ptr[b*sB + t*sT + f*sF] (float units), nullptr = only write crm
    long sB, sT, sF;
    cf2 *out;           // (b, t, f) at b*oB + t*oT + f*oF
    long oB, oT, oF;
    float *crm;         // optional copy of the compressed mask in the reference layout [B][2][F][T], nullptr = off
    int T, F;
};

void launch_k_fsn_mask(dim3 grid, hipStream_t st, const FsnMaskArgs &a);  // defined in se_aux.hip

#ifdef SE_AUX_KERNELS
__global__ void k_fsn_mask(FsnMaskArgs a) {
    const int b = blockIdx.y;
    const int TF = a.T * a.F;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= TF) return;
    const int f = i / a.T, t = i - f * a.T;
    const float cr = a.mask[(((long)b * a.F + f) * 2 + 0) * a.T + t], ci = a.mask[(((long)b * a.F + f) * 2 + 1) * a.T + t];
    if (a.crm) {
        a.crm[(((long)b * 2 + 0) * a.F + f) * a.T + t] = cr;
        a.crm[(((long)b * 2 + 1) * a.F + f) * a.T + t] = ci;
    }
    if (!a.re) return;
    const float mr = decompress_cirm(cr), mi = decompress_cirm(ci);
    const long off = (long)b * a.sB + (long)t * a.sT + (long)f * a.sF;
    const float nx = a.re[off], ny = a.im[off];
    a.out[(long)b * a.oB + (long)t * a.oT + (long)f * a.oF] = cf2{mr * nx - mi * ny, mi * nx + mr * ny};
}

#endif  // SE_AUX_KERNELS

}  // namespace se
