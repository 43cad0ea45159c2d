// 16-bit storage / matrix-core helpers shared by the mixed-precision kernels (gfx950 only).
//
// PREC (template parameter of the matrix-core kernels, = dsg_dtype of dsg_conv_args.compute_dtype):
//   0  fp32-equivalent: fp32 tensors, every product as an fp16x2 split (3 f16 MFMAs)
//   1  bf16: channel-blocked tensors are bf16 in HBM, one v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate
//   2  fp16: the same with _Float16 / v_mfma_f32_32x32x16_f16 (the reference's train.py:24 mixed_precision='fp16')
// [N,C,H,W] tensors (the public boundary, the attention kernel's q/k/v) stay fp32 in every mode.
#pragma once
#include "dsg_common.h"

namespace dsg {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// D += A * B on the 16-bit matrix cores; operands travel as 8 x 16 bits in a half8 container
template <int PREC>
__device__ __forceinline__ f32x16 mma16(half8 a, half8 b, f32x16 c) {
  if constexpr (PREC == 1)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// two fp32 -> one dword holding (lo = a, hi = b) in the 16-bit type of PREC (round to nearest even)
template <int PREC>
__device__ __forceinline__ unsigned pack2(float a, float b) {
  if constexpr (PREC == 1) {
    const bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
  } else {
    const half2v v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, v);
  }
}
template <int PREC>
__device__ __forceinline__ float lo16(unsigned w) {
  if constexpr (PREC == 1) return __uint_as_float(w << 16);
  else return (float)__builtin_bit_cast(half2v, w)[0];
}
template <int PREC>
__device__ __forceinline__ float hi16(unsigned w) {
  if constexpr (PREC == 1) return __uint_as_float(w & 0xFFFF0000u);
  else return (float)__builtin_bit_cast(half2v, w)[1];
}
// one 16-bit value (runtime dtype: 1 bf16, 2 fp16) <-> fp32, for the streaming kernels where the type is an argument
__device__ __forceinline__ float ld16(const unsigned short* p, int dt) {
  const unsigned short h = *p;
  return dt == 1 ? __uint_as_float((unsigned)h << 16) : (float)__builtin_bit_cast(_Float16, h);
}
__device__ __forceinline__ unsigned short cvt16(float v, int dt) {
  if (dt == 1) return __builtin_bit_cast(unsigned short, (__bf16)v);
  return __builtin_bit_cast(unsigned short, (_Float16)v);
}
__device__ __forceinline__ float word_lo(unsigned w, int dt) { return dt == 1 ? lo16<1>(w) : lo16<2>(w); }
__device__ __forceinline__ float word_hi(unsigned w, int dt) { return dt == 1 ? hi16<1>(w) : hi16<2>(w); }
__device__ __forceinline__ unsigned word_pack(float a, float b, int dt) { return dt == 1 ? pack2<1>(a, b) : pack2<2>(a, b); }

}  // namespace dsg
