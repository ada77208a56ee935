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
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// native conversions: gfx950 has v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 (round to nearest even, 2 values per
// instruction).  A software RNE costs ~6 VALU per element and made the attention kernels VALU-bound.
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)f);
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
template <int DT> __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  if constexpr (DT == MH_BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
  else return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
// raw v_exp_f32 (no denormal-range fix-up code): softmax arguments are <= 0, tiny results may flush to 0
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
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

// rotate-half RoPE of one (x[i], x[i + D/2]) pair; ONE definition with explicit fmas so that the stand-alone kernel, the
// decode kernel and the GEMM-epilogue form round identically (inverse: pass -sin)
// The fp32 results pass through an empty asm: without it hipcc may merge the fma with the caller's 16-bit store conversion into ONE
// v_fma_mix*_f16 (a single rounding from the exact value) in some callers and not in others, and the kernels then disagree wherever
// the fp32 result is a 16-bit tie (found by tests/test_ops_gpu.py: fused q|k|v epilogue vs mh_decode_rope_append).
__device__ __forceinline__ float fp32_value(float x) {
  asm("" : "+v"(x));
  return x;
}
__device__ __forceinline__ void rope_rot(float a, float b, float c, float s, float& lo, float& hi) {
  const float l = fmaf(a, c, -(b * s)), h = fmaf(b, c, a * s);
  lo = fp32_value(l);
  hi = fp32_value(h);
}

// SwiGLU element maths, ONE definition (explicit fma) shared by the stand-alone kernels and the GEMM-epilogue forms so that
// both round identically: act = silu(g) * u;  d_up = d * g * sig(g);  d_gate = d * u * sig(g) * (1 + g * (1 - sig(g)))
// (v_rcp_f32, 1 ulp, instead of the IEEE division sequence hipcc emits for `1.0f / x` - ~10 instructions per element, 128 elements per lane in a
//  SwiGLU store phase; one definition for every form, so fused and unfused kernels still agree bit for bit)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float swiglu_fwd1(float g, float u) { return fp32_value(g * sigmoidf_(g) * u); }
__device__ __forceinline__ void swiglu_bwd1(float g, float u, float d, float& dg, float& du) {
  const float s = sigmoidf_(g);
  du = d * g * s;
  dg = d * u * s * fmaf(g, 1.0f - s, 1.0f);
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

// ---- hand-waited LDS reads ---------------------------------------------------------------------------
// hipcc puts `s_waitcnt vmcnt(0)` in front of every ds_read it can see after a global_load_lds (it cannot
// prove the read does not alias the in-flight LDS-DMA), which turns a prefetch into a synchronous load.
// Fragment reads that must overlap an in-flight stage are therefore inline asm (invisible to that pass) and
// their completion is waited for by hand: LGKM_WAIT(n) = s_waitcnt lgkmcnt(n) + sched_barrier (rule 18:
// the MFMAs that consume the registers must not be hoisted above the wait).
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void lds_read128(u32x4_t& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
#define LGKM_WAIT(N)                                             \
  do {                                                           \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory");      \
    __builtin_amdgcn_sched_barrier(0);                           \
  } while (0)
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
template <int DT> __device__ __forceinline__ f32x16_t mfma32v(u32x4_t a, u32x4_t b, f32x16_t c) {
  if constexpr (DT == MH_BF16)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
template <int DT> __device__ __forceinline__ u32x4_t pack8v(const float* f) {
  u32x4_t v;
  v[0] = pack2<DT>(f[0], f[1]); v[1] = pack2<DT>(f[2], f[3]);
  v[2] = pack2<DT>(f[4], f[5]); v[3] = pack2<DT>(f[6], f[7]);
  return v;
}

// async global -> LDS, 4 bytes per lane (LDS destination = wave-uniform base + lane*4)
__device__ __forceinline__ void glds4(const void* gsrc, void* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_base, 4, 0, 0);
}
// async global -> LDS, 16 bytes per lane; LDS destination = wave-uniform `lds_base` + lane*16
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

// XCD-aware 1-D grid for (batch*head, chunk) work: block b runs on XCD b%8 (observed dispatch rule, used for
// speed only).  All `nchunk` blocks of one (b,h) are given to ONE XCD, consecutively, so the K/V (or Q/dO)
// panels they share stay in that XCD's private L2.  Grid = round_up(BH, 8) * nchunk; returns false for the
// padding blocks.  chunk order is the dispatch order within the (b,h) (causal kernels put their longest chunk first).
// The LAST `XCD_TAIL` (b,h) of every XCD are dispatched chunk-major instead (chunk 0 of each of them, then chunk 1, ...):
// causal chunks differ up to 32:1 in length, and a long chunk that starts when the queue is nearly empty leaves the
// other CUs of the XCD idle for most of its run.
#ifndef MH_XCD_TAIL
#define MH_XCD_TAIL 4
#endif
constexpr int XCD_TAIL = MH_XCD_TAIL;
__device__ __forceinline__ bool xcd_work(int BH, int nchunk, int& bh, int& chunk) {
  const int id = blockIdx.x;
  const int xcd = id & 7, idx = id >> 3;
  const int nb = (int)(gridDim.x >> 3) / nchunk;  // (b,h) per XCD
  const int tail = nb < XCD_TAIL ? nb : XCD_TAIL;
  const int head_blocks = (nb - tail) * nchunk;
  int bl;
  if (idx < head_blocks) {
    bl = idx / nchunk;
    chunk = idx % nchunk;
  } else {
    const int r = idx - head_blocks;
    chunk = r / tail;
    bl = (nb - tail) + r % tail;
  }
  bh = bl * 8 + xcd;
  return bh < BH;
}
static inline int xcd_grid(int BH, int nchunk) { return ((BH + 7) / 8) * 8 * nchunk; }

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
