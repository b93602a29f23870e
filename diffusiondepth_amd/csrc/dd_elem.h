// dd_elem.h -- element-kind helpers shared by the fused convolution kernels (dd_igemm2.hip) and the backward kernels:
// bf16/f16/f32 <-> fp32 conversion of 16-byte pieces, the MFMA step, and the channel-blocked activation layout.
#pragma once
#include "dd_kernels.h"
#ifndef DD_NO_PK_F32
#define DD_NO_PK_F32 1     // scalar FMAs in the GroupNorm prologue (0 = v_pk_fma_f32: slower beside MFMAs)
#endif

namespace dd {

// Activations with C >= 32 channels are stored channel-blocked: [B][C/32][H][W][32] elements, so
// that one 32-channel block of a row of pixels is contiguous (64 B per pixel in bf16/f16, 128 B in
// fp32): conv epilogues write and conv prologues read whole contiguous row segments per block.
// 16-channel tensors (state x, conv4 output) are plain NHWC [B][H][W][16].
constexpr int ACT_CB = 32;
__host__ __device__ inline size_t act_offset(int C, int h, int w, int b, int c, int y, int x) {
  if (C < ACT_CB) return (((size_t)b * h + y) * w + x) * C + c;
  return ((((size_t)b * (C / ACT_CB) + c / ACT_CB) * h + y) * w + x) * ACT_CB + (c % ACT_CB);
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// ------------------------------------------------------------------------------------------------
// element helpers
// ------------------------------------------------------------------------------------------------
template <int EK> struct ElemSize { static constexpr int V = (EK == EK_F32) ? 4 : 2; };

__device__ __forceinline__ float bf16_to_f32(uint32_t u16) { return __builtin_bit_cast(float, u16 << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16(float f) {            // round-to-nearest-even
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ float f16_to_f32(uint32_t u16) {
  return (float)__builtin_bit_cast(_Float16, (uint16_t)u16);
}
__device__ __forceinline__ uint32_t f32_to_f16(float f) {
  return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)f);
}

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
// two fp32 -> one packed dword, round-to-nearest-even in hardware (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32)
template <int EK> __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  if constexpr (EK == EK_BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
  else return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}

// 16-byte piece <-> floats.  NE = elements per piece (8 for 2-byte kinds, 4 for fp32).
template <int EK> struct Piece {
  static constexpr int NE = 16 / ElemSize<EK>::V;
  static __device__ __forceinline__ void unpack(const uint4& v, float (&f)[NE]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    if constexpr (EK == EK_F32) {
#pragma unroll
      for (int i = 0; i < 4; ++i) f[i] = __builtin_bit_cast(float, w[i]);
    } else if constexpr (EK == EK_BF16) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { f[2 * i] = bf16_to_f32(w[i] & 0xFFFFu); f[2 * i + 1] = bf16_to_f32(w[i] >> 16); }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) { f[2 * i] = f16_to_f32(w[i] & 0xFFFFu); f[2 * i + 1] = f16_to_f32(w[i] >> 16); }
    }
  }
  static __device__ __forceinline__ uint4 pack(const float (&f)[NE]) {
    uint32_t w[4];
    if constexpr (EK == EK_F32) {
#pragma unroll
      for (int i = 0; i < 4; ++i) w[i] = __builtin_bit_cast(uint32_t, f[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) w[i] = pack2<EK>(f[2 * i], f[2 * i + 1]);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
};

// relu(a*y + b) of one 16-byte piece of 2-byte elements, result packed again (the GroupNorm + ReLU prologue when nothing
// is added behind the ReLU).  VALU diet: packed fp32 FMA (v_pk_fma_f32), hardware RNE pack, and the ReLU on the PACKED
// pair as a signed 16-bit integer max with 0 (v_pk_max_i16: negative bf16/f16 values have the sign bit set, and rounding
// commutes with the ReLU) -- 20 VALU per 8 elements instead of 28.
typedef __attribute__((ext_vector_type(2))) short i16x2_t;
// INK = kind of the stored input, EK = kind of the packed result (the MFMA operand)
template <int INK, int EK>
__device__ __forceinline__ uint4 affine_relu_pack(const uint4& raw, const float (&ta)[8], const float (&tb)[8]) {
  static_assert(EK != EK_F32 && INK != EK_F32, "2-byte element kinds only");
  const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
  uint32_t o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x2_t y;
    if constexpr (INK == EK_BF16) { y.x = __builtin_bit_cast(float, w[i] << 16); y.y = __builtin_bit_cast(float, w[i] & 0xFFFF0000u); }
    else { y.x = f16_to_f32(w[i] & 0xFFFFu); y.y = f16_to_f32(w[i] >> 16); }
#if DD_NO_PK_F32
    f32x2_t u;                       // two scalar FMAs: packed fp32 VALU beside running MFMAs costs ~22 cycles more per instruction (MI355X_MICROARCH.md)
    u.x = __builtin_fmaf(ta[2 * i], y.x, tb[2 * i]);
    u.y = __builtin_fmaf(ta[2 * i + 1], y.y, tb[2 * i + 1]);
#else
    const f32x2_t a = {ta[2 * i], ta[2 * i + 1]}, b = {tb[2 * i], tb[2 * i + 1]};
    const f32x2_t u = __builtin_elementwise_fma(a, y, b);
#endif
    const i16x2_t pk = __builtin_bit_cast(i16x2_t, pack2<EK>(u.x, u.y));
    const i16x2_t zero = {0, 0};
    o[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(pk, zero));
  }
  return make_uint4(o[0], o[1], o[2], o[3]);
}

template <int EK>
__device__ __forceinline__ void mma_step(f32x16_t& acc, const uint4& wf, const uint4& pf) {
  if constexpr (EK == EK_BF16) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf), __builtin_bit_cast(bf16x8_t, pf), acc, 0, 0, 0);
  } else if constexpr (EK == EK_F16) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, wf), __builtin_bit_cast(f16x8_t, pf), acc, 0, 0, 0);
  } else {
    // lane (i, g) holds channels 4g..4g+3 of an 8-channel group: MFMA j contracts channels {j, 4+j}
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, wf.x), __builtin_bit_cast(float, pf.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, wf.y), __builtin_bit_cast(float, pf.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, wf.z), __builtin_bit_cast(float, pf.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, wf.w), __builtin_bit_cast(float, pf.w), acc, 0, 0, 0);
  }
}

}  // namespace dd
