// se_convp_dispatch.hip - routes conv_p_launch() to the per-(planes, taps) translation units of se_convp.hip.
#include <hip/hip_runtime.h>

#define SE_NO_NORM_KERNELS 1
#include "convp_dispatch.h"

namespace se {

#define SE_DECL(PL_, NT_)                                                                                                  \
    int conv_p_launch_pl##PL_##_t##NT_(int NT, int CO, dim3 grid, size_t lds, hipStream_t st, const ConvPArgs &a); \
    void conv_p_set_attributes_pl##PL_##_t##NT_();
#define SE_DECL3(NT_) SE_DECL(1, NT_) SE_DECL(2, NT_) SE_DECL(3, NT_)
SE_DECL3(25) SE_DECL3(15) SE_DECL3(9) SE_DECL3(6) SE_DECL3(1)

int conv_p_launch(int ntap, int NT, int CO, int PL, dim3 grid, size_t lds, hipStream_t st, const ConvPArgs &a) {
#define SE_ROUTE(PL_, NT_) \
    if (PL == PL_ && ntap == NT_) return conv_p_launch_pl##PL_##_t##NT_(NT, CO, grid, lds, st, a);
#define SE_ROUTE3(NT_) SE_ROUTE(1, NT_) SE_ROUTE(2, NT_) SE_ROUTE(3, NT_)
    SE_ROUTE3(25) SE_ROUTE3(15) SE_ROUTE3(9) SE_ROUTE3(6) SE_ROUTE3(1)
    return 1;
}

bool conv_p_has_instance(int ntap, int NT, int CO) {
    const bool nt_ok = NT == 1 || NT == 2 || NT == 3 || NT == 4 || NT == 6 || NT == 8 || NT == 10 || NT == 12;
    if (!nt_ok) return false;
    if (ntap == 1) return CO == 1 || CO == 2 || CO == 4;
    if (ntap == 25) return CO == 1 && NT <= 4;
    return (ntap == 15 || ntap == 9 || ntap == 6) && CO == 1;
}

void conv_p_set_attributes() {
#define SE_ATTR(PL_, NT_) conv_p_set_attributes_pl##PL_##_t##NT_();
#define SE_ATTR3(NT_) SE_ATTR(1, NT_) SE_ATTR(2, NT_) SE_ATTR(3, NT_)
    SE_ATTR3(25) SE_ATTR3(15) SE_ATTR3(9) SE_ATTR3(6) SE_ATTR3(1)
}

}  // namespace se
