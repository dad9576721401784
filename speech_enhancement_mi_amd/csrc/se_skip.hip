// se_skip.hip - translation unit of the streaming skip-gate kernel (skip_p.hip.h).
#include <hip/hip_runtime.h>

#define SE_NO_NORM_KERNELS 1
#include "convp_dispatch.h"
#include "skip_p.hip.h"

namespace se {

int launch_k_skip_p(int PL, int KP, int batch, size_t lds, hipStream_t st, const SkipPArgs &a) {
#define SE_SK(PL_, KP_) \
    if (PL == PL_ && KP == KP_) { hipLaunchKernelGGL((k_skip_p<PL_, KP_, (KP_ * PL_ <= 4 ? 16 : 8)>), dim3(batch), dim3(KP_ * PL_ <= 4 ? 1024 : 512), lds, st, a); return 0; }
#define SE_SK3(KP_) SE_SK(1, KP_) SE_SK(2, KP_) SE_SK(3, KP_)
    SE_SK3(1) SE_SK3(2) SE_SK3(4)
    return 1;
}

void skip_p_set_attributes() {
#define SE_SKA(PL_, KP_) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_skip_p<PL_, KP_, (KP_ * PL_ <= 4 ? 16 : 8)>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
#define SE_SKA3(KP_) SE_SKA(1, KP_) SE_SKA(2, KP_) SE_SKA(3, KP_)
    SE_SKA3(1) SE_SKA3(2) SE_SKA3(4)
}

}  // namespace se
