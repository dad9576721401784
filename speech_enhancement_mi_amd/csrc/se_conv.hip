// se_conv.hip - translation unit of the convolution kernels (k_conv_x6 / k_conv_igemm / k_conv_small template instances) and
// their host-side dispatch (conv_dispatch.h).  Built with -fno-slp-vectorize like every TU of the library (see se_aux.hip).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define SE_NO_NORM_KERNELS 1
#include "conv_dispatch.h"
#include "conv_igemm.hip.h"

namespace se {

int conv_x6_launch_pl1(int ntap, int NT, int CO, dim3 grid, size_t lds, hipStream_t st, const ConvX6Args &xa);
int conv_x6_launch_pl2(int ntap, int NT, int CO, dim3 grid, size_t lds, hipStream_t st, const ConvX6Args &xa);
int conv_x6_launch_pl3(int ntap, int NT, int CO, dim3 grid, size_t lds, hipStream_t st, const ConvX6Args &xa);
void conv_x6_set_attributes_pl1();
void conv_x6_set_attributes_pl2();
void conv_x6_set_attributes_pl3();

int conv_x6_launch(int ntap, int NT, int CO, int PL, dim3 grid, size_t lds, hipStream_t st, const ConvX6Args &xa) {
    return PL == 1 ? conv_x6_launch_pl1(ntap, NT, CO, grid, lds, st, xa)
                   : (PL == 2 ? conv_x6_launch_pl2(ntap, NT, CO, grid, lds, st, xa) : conv_x6_launch_pl3(ntap, NT, CO, grid, lds, st, xa));
}

int conv_igemm_launch(int ntap, int NT, int CoPad, dim3 grid, size_t lds, hipStream_t st, const ConvArgs &a) {
#define SE_CONV_CASE(NTAP_, NT_) \
    case NTAP_ * 8 + NT_: hipLaunchKernelGGL((k_conv_igemm<NTAP_, NT_>), grid, dim3(256), lds, st, a); return 0;
#define SE_CONV_TAPS(NTAP_) SE_CONV_CASE(NTAP_, 1) SE_CONV_CASE(NTAP_, 2) SE_CONV_CASE(NTAP_, 3) SE_CONV_CASE(NTAP_, 4) \
    case NTAP_ * 8: if (CoPad == 4) hipLaunchKernelGGL((k_conv_small<NTAP_, 1>), grid, dim3(256), lds, st, a); \
                    else if (CoPad == 8) hipLaunchKernelGGL((k_conv_small<NTAP_, 2>), grid, dim3(256), lds, st, a); \
                    else hipLaunchKernelGGL((k_conv_small<NTAP_, 4>), grid, dim3(256), lds, st, a); return 0;
    switch (ntap * 8 + NT) {
        SE_CONV_TAPS(15) SE_CONV_TAPS(9) SE_CONV_TAPS(6) SE_CONV_TAPS(1)
        case 25 * 8: {
            // the pre-conv blocks in their reference geometry (5 channels = 3 microphones, stride 1, time taps t-4..t, the fused
            // gated pair, all channels activated): two time rows per thread; anything else on the general small kernel
            bool std5 = a.Co == 5 && a.Ci == 5 && CoPad == 8 && a.s == 1 && a.os == 1 && a.dil == 1 && a.tlo_off == -4 && a.gatew != nullptr &&
                        a.relu_lo == 0 && a.relu_hi >= 5 && a.Fi == a.FP;
            for (int k = 0; k < 25 && std5; k++) std5 = a.rowgrp[k] == k % 5 && a.coloff[k] == a.coloff[(k / 5) * 5];
            if (std5) hipLaunchKernelGGL((k_preconv_tb<5>), grid, dim3(256), lds, st, a);
            else hipLaunchKernelGGL((k_conv_small<25, 2>), grid, dim3(256), lds, st, a);
            return 0;
        }
        default: return 1;
    }
#undef SE_CONV_TAPS
#undef SE_CONV_CASE
}

void conv_set_attributes() {
    const int kMax = 160 * 1024;
#define SE_CONV_ATTR(NTAP_)                                                                                                        \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv_igemm<NTAP_, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax); \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv_igemm<NTAP_, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax); \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv_igemm<NTAP_, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax); \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv_igemm<NTAP_, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax); \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv_small<NTAP_, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax);
    SE_CONV_ATTR(15) SE_CONV_ATTR(9) SE_CONV_ATTR(6) SE_CONV_ATTR(1)
#undef SE_CONV_ATTR
    conv_x6_set_attributes_pl1();
    conv_x6_set_attributes_pl2();
    conv_x6_set_attributes_pl3();
}

}  // namespace se
