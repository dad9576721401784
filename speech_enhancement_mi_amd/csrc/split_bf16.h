// split_bf16.h - the operand split behind the "bf16x6 / bf16x3" contractions: an fp32 value is the exact sum of three bf16 planes
// x = hi + mid + lo (8 + 8 + 8 mantissa bits; every difference below is exact in fp32), see gemm.hip.h.
#pragma once
#include <hip/hip_runtime.h>

namespace se {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float x, __bf16 &h, __bf16 &m, __bf16 &l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

}  // namespace se
