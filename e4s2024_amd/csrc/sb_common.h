// Split-bf16 helpers shared by the modulated-conv kernels (modconv_sb.hip, modconv_upfused.hip).
#pragma once
#include "common.h"

namespace e4s {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int CKS = 16;  // input channels per K chunk (= one 16-deep MFMA step per tap)

__device__ __forceinline__ unsigned pack_bf16_rne(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));  // v_cvt_pk_bf16_f32
}
// (t0, t1) -> packed hi pair, packed lo pair
__device__ __forceinline__ void split2(float t0, float t1, unsigned& hi, unsigned& lo) {
    hi = pack_bf16_rne(t0, t1);
    const float h0 = __builtin_bit_cast(float, hi << 16);
    const float h1 = __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = pack_bf16_rne(t0 - h0, t1 - h1);
}

}  // namespace e4s
