// Shared device/host helpers for libmerlin_hip.so (gfx950 only; no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/merlin_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define MH_LAUNCH_CHECK()                         \
  do {                                            \
    hipError_t e_ = hipGetLastError();            \
    if (e_ != hipSuccess) return (int)e_;         \
    return MH_OK;                                 \
  } while (0)

// ---- 16-bit storage types: tag-dispatched conversions (DT = MH_BF16 / MH_F16) ---------------
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0u;  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);                       // round to nearest even
  return u >> 16;
}
__device__ __forceinline__ float f16_bits_to_f32(uint32_t h) {
  uint16_t s = (uint16_t)h;
  return (float)__builtin_bit_cast(_Float16, s);
}
__device__ __forceinline__ uint32_t f32_to_f16_bits(float f) {
  _Float16 h = (_Float16)f;
  return (uint32_t)__builtin_bit_cast(uint16_t, h);
}
template <int DT> __device__ __forceinline__ float ld16(uint32_t bits) {
  if constexpr (DT == MH_BF16) return bf16_bits_to_f32(bits & 0xffffu);
  else return f16_bits_to_f32(bits & 0xffffu);
}
template <int DT> __device__ __forceinline__ uint32_t st16(float f) {
  if constexpr (DT == MH_BF16) return f32_to_bf16_bits(f);
  else return f32_to_f16_bits(f);
}
template <int DT> __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return st16<DT>(lo) | (st16<DT>(hi) << 16); }
template <int DT> __device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi) {
  lo = ld16<DT>(w & 0xffffu);
  hi = ld16<DT>(w >> 16);
}
// 8 x 16-bit <-> 8 x fp32
template <int DT> __device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  unpack2<DT>(v.x, f[0], f[1]); unpack2<DT>(v.y, f[2], f[3]);
  unpack2<DT>(v.z, f[4], f[5]); unpack2<DT>(v.w, f[6], f[7]);
}
template <int DT> __device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack2<DT>(f[0], f[1]); v.y = pack2<DT>(f[2], f[3]);
  v.z = pack2<DT>(f[4], f[5]); v.w = pack2<DT>(f[6], f[7]);
  return v;
}

// ---- MFMA wrappers ------------------------------------------------------------------------
// 16x16x32: A lane l holds A[i=l&15][k=8*(l>>4)..+8]; B lane l holds B[k=8*(l>>4)..+8][j=l&15];
// D lane l reg r holds D[i=4*(l>>4)+r][j=l&15].
template <int DT> __device__ __forceinline__ f32x4_t mfma16(const uint4& a, const uint4& b, f32x4_t c) {
  if constexpr (DT == MH_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
// 32x32x16: A lane l holds A[i=l&31][k=8*(l>>5)..+8]; B lane l holds B[k=8*(l>>5)..+8][j=l&31];
// D lane l reg r holds D[i=(r&3)+8*(r>>2)+4*(l>>5)][j=l&31].
template <int DT> __device__ __forceinline__ f32x16_t mfma32(const uint4& a, const uint4& b, f32x16_t c) {
  if constexpr (DT == MH_BF16)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// ---- wave reductions (wave = 64 lanes) ------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// async global -> LDS, 16 bytes per lane; LDS destination = wave-uniform `lds_base` + lane*16
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
