// se_conv_x6.hip - k_conv_x6 template instances for ONE operand-plane count (compiled three times: -DSE_X6_PL=1, 2, 3 ->
// se_conv_x6_pl{1,2,3}.o) so that the 72 instances build in parallel.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define SE_NO_NORM_KERNELS 1
#include "conv_dispatch.h"
#include "conv_x6.hip.h"

#ifndef SE_X6_PL
#error "compile with -DSE_X6_PL=1|2|3"
#endif
#define SE_CAT2(a, b) a##b
#define SE_CAT(a, b) SE_CAT2(a, b)

namespace se {

int SE_CAT(conv_x6_launch_pl, SE_X6_PL)(int ntap, int NT, int CO, dim3 grid, size_t lds, hipStream_t st, const ConvX6Args &xa) {
#define SE_X6_CASE(NTAP_, NT_, CO_) \
    case (NTAP_ * 16 + NT_) * 8 + CO_: hipLaunchKernelGGL((k_conv_x6<NTAP_, NT_, CO_, SE_X6_PL>), grid, dim3(256), lds, st, xa); return 0;
#define SE_X6_TAPS(NTAP_, CO_) SE_X6_CASE(NTAP_, 1, CO_) SE_X6_CASE(NTAP_, 2, CO_) SE_X6_CASE(NTAP_, 3, CO_) SE_X6_CASE(NTAP_, 4, CO_)
    switch ((ntap * 16 + NT) * 8 + CO) {
        SE_X6_TAPS(15, 1) SE_X6_TAPS(9, 1) SE_X6_TAPS(6, 1) SE_X6_TAPS(1, 1) SE_X6_TAPS(1, 2) SE_X6_TAPS(1, 4)
        default: return 1;
    }
#undef SE_X6_TAPS
#undef SE_X6_CASE
}

void SE_CAT(conv_x6_set_attributes_pl, SE_X6_PL)() {
    const int kMax = 160 * 1024;
#define SE_X6_ATTR1(NTAP_, NT_, CO_) \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv_x6<NTAP_, NT_, CO_, SE_X6_PL>), hipFuncAttributeMaxDynamicSharedMemorySize, kMax);
#define SE_X6_ATTR(NTAP_, CO_) SE_X6_ATTR1(NTAP_, 1, CO_) SE_X6_ATTR1(NTAP_, 2, CO_) SE_X6_ATTR1(NTAP_, 3, CO_) SE_X6_ATTR1(NTAP_, 4, CO_)
    SE_X6_ATTR(15, 1) SE_X6_ATTR(9, 1) SE_X6_ATTR(6, 1) SE_X6_ATTR(1, 1) SE_X6_ATTR(1, 2) SE_X6_ATTR(1, 4)
#undef SE_X6_ATTR
#undef SE_X6_ATTR1
}

#if SE_X6_PL == 3
void conv_x6_trace_dump() {
#ifdef SE_X6_TRACE
    unsigned long long t[16] = {0};
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_x6_trace), sizeof(t));
    if (t[15]) {
        const char *nm[9] = {"prologue + pair loops", "barrier1", "wait staging loads", "split+lds write", "barrier2", "first A frag", "epilogue issue", "epilogue store drain", "stats reduce"};
        fprintf(stderr, "[x6 trace %s] %llu WG-samples\n", getenv("SE_X6_TRACE_LABEL"), t[15]);
        for (int i = 0; i < 9; i++) fprintf(stderr, "   %-28s %10.0f cycles/WG\n", nm[i], (double)t[i] / t[15]);
    }
#endif
}
#endif

}  // namespace se
