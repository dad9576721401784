// se_convp.hip - k_conv_p template instances for ONE (operand planes, tap count) pair: compiled 12 times
// (-DSE_CP_PL=1|2|3 -DSE_CP_NTAP=15|9|6|1 -> se_convp_pl<P>_t<N>.o) so that the instances build in parallel.
// The (PL = 3, NTAP = 15) unit also carries the small producer / consumer kernels of the two layouts.
#include <hip/hip_runtime.h>

#define SE_NO_NORM_KERNELS 1
#include "convp_dispatch.h"
#include "conv_p.hip.h"

#if !defined(SE_CP_PL) || !defined(SE_CP_NTAP)
#error "compile with -DSE_CP_PL=1|2|3 -DSE_CP_NTAP=25|15|9|6|1"
#endif
#define SE_CAT4_(a, b, c, d) a##b##c##d
#define SE_CAT4(a, b, c, d) SE_CAT4_(a, b, c, d)
#define SE_FN(name) SE_CAT4(name, SE_CP_PL, _t, SE_CP_NTAP)

namespace se {

// tiles per wave that exist as instances; multi-tap convolutions stage one channel octet per chunk, 1x1 convolutions 2 or 4
#define SE_CP_NTS(X, CO_) X(1, CO_) X(2, CO_) X(3, CO_) X(4, CO_) X(6, CO_) X(8, CO_) X(10, CO_) X(12, CO_)
#if SE_CP_NTAP == 1
#define SE_CP_ALL(X) SE_CP_NTS(X, 1) SE_CP_NTS(X, 2) SE_CP_NTS(X, 4)
#elif SE_CP_NTAP == 25  // 5x5 pre-conv blocks of CRN_ELU: one channel octet, patches of <= 4 tiles per wave fit the LDS-DMA budget
#define SE_CP_ALL(X) X(1, 1) X(2, 1) X(3, 1) X(4, 1)
#else
#define SE_CP_ALL(X) SE_CP_NTS(X, 1)
#endif

int SE_FN(conv_p_launch_pl)(int NT, int CO, dim3 grid, size_t lds, hipStream_t st, const ConvPArgs &a) {
#define SE_CP_CASE(NT_, CO_) \
    case NT_ * 8 + CO_: hipLaunchKernelGGL((k_conv_p<SE_CP_NTAP, NT_, CO_, SE_CP_PL>), grid, dim3(256), lds, st, a); return 0;
    switch (NT * 8 + CO) {
        SE_CP_ALL(SE_CP_CASE)
        default: return 1;
    }
#undef SE_CP_CASE
}

void SE_FN(conv_p_set_attributes_pl)() {
#define SE_CP_ATTR(NT_, CO_) \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv_p<SE_CP_NTAP, NT_, CO_, SE_CP_PL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    SE_CP_ALL(SE_CP_ATTR)
#undef SE_CP_ATTR
}

#if SE_CP_PL == 3 && SE_CP_NTAP == 15
void launch_k_featurize_p(int PL, dim3 grid, hipStream_t st, const FeatPArgs &a) {
    if (PL == 1) hipLaunchKernelGGL(k_featurize_p<1>, grid, dim3(256), 0, st, a);
    else if (PL == 2) hipLaunchKernelGGL(k_featurize_p<2>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_featurize_p<3>, grid, dim3(256), 0, st, a);
}
void launch_k_f32_to_p(int PL, dim3 grid, hipStream_t st, const F32ToPArgs &a) {
    if (PL == 1) hipLaunchKernelGGL(k_f32_to_p<1>, grid, dim3(256), 0, st, a);
    else if (PL == 2) hipLaunchKernelGGL(k_f32_to_p<2>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_f32_to_p<3>, grid, dim3(256), 0, st, a);
}
void launch_k_gln_p(int PL, dim3 grid, hipStream_t st, const GlnPArgs &a) {
    if (PL == 1) hipLaunchKernelGGL(k_gln_p<1>, grid, dim3(256), 0, st, a);
    else if (PL == 2) hipLaunchKernelGGL(k_gln_p<2>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_gln_p<3>, grid, dim3(256), 0, st, a);
}
void launch_k_gln2_p(int PL, dim3 grid, hipStream_t st, const Gln2PArgs &a) {
    if (PL == 1) hipLaunchKernelGGL(k_gln2_p<1>, grid, dim3(1024), 0, st, a);
    else if (PL == 2) hipLaunchKernelGGL(k_gln2_p<2>, grid, dim3(1024), 0, st, a);
    else hipLaunchKernelGGL(k_gln2_p<3>, grid, dim3(1024), 0, st, a);
}
void launch_k_final_mask_p(dim3 grid, hipStream_t st, const MaskPArgs &a) { hipLaunchKernelGGL(k_final_mask_p<0>, grid, dim3(256), 0, st, a); }
#endif

}  // namespace se
