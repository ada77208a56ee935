// Shared declarations of the attention-backward kernels (see attn_bwd.hip for the design notes).
#pragma once
#include "mh_common.h"
#include "merlin_hip_dev.h"

namespace mhattn {

struct BwdArgs {
  const uint16_t *q, *k, *v, *o, *dout;
  const uint16_t *qt, *dot, *kt;  // [B, H, D, S_pad] permuted transposes
  const float* lse;
  float* delta;        // [2][B, H, S_pad]: delta = rowsum(dO o O), then lse2 = lse * log2(e)
  uint16_t *dq, *dk, *dv;
  const int32_t* seqlens;
  int64_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  int B, S, H, S_pad;
  float scale, scale_log2;
  const float* lse2;   // = delta + B*H*S_pad (filled by delta_k)
};

template <int D> struct RSwz;  // swizzle of a [rows][D] 16-bit tile (row = D*2 bytes)
template <> struct RSwz<128> { static __device__ __forceinline__ int f(int row) { return row & 15; } };
template <> struct RSwz<64> { static __device__ __forceinline__ int f(int row) { return (row >> 1) & 7; } };
// swizzle of a [rows][32] 16-bit tile (64-byte rows, 4 chunks): 4 rows share a 256-B bank row
__device__ __forceinline__ int tswz(int row) { return (row >> 2) & 3; }

// defined in attn_bwd_kv.hip: dV and dK kernels (KV-block outer loop)
int launch_attn_bwd_kv(const BwdArgs& a, int dt, int D, int causal, hipStream_t stream);

}  // namespace mhattn
