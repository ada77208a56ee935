// Shared pieces of the MFMA GEMM kernels (see gemm.hip for the design notes).
#pragma once
#include "mh_common.h"

namespace mhgemm {

struct GemmArgs {
  const uint16_t* A;
  const uint16_t* B;
  void* C;
  const uint16_t* bias;
  const uint16_t* resid;
  int64_t lda, ldb, ldc, ldr;
  int M, N, K, epi, tiles_m, tiles_n, vec_ok;
  int splits;              // split-K: grid.y blocks share a tile, block y writes fp32 partials to C + y*c_split
  int64_t c_split;         // elements between the partial outputs of consecutive splits
  const uint16_t* zero_row;  // >= 1 KiB of zeros: source of K-strided rows k >= K in the last K-tile
  // fused RoPE of the staged epilogue (qkv projection): columns [0, rope_cols) are heads of rope_D channels rotated
  // at position (row % rope_S) with the (cos, sin) table rope_tab [pos][rope_D/2]; nullptr = off
  const float* rope_tab;
  int rope_S, rope_D, rope_cols;
  // fused SwiGLU of the staged epilogue.  sw_mode 1 (forward, gate|up projection): a tile is 128 gate columns [n0, n0+128)
  // + the 128 up columns ff + [n0, n0+128) (B rows taken from both halves of the fused weight); C = gu [M, 2 ff] is
  // written as usual and sw_out = act [M, ff] = silu(gate) * up.  sw_mode 2 (backward, dact = dY Wd): the tile of dact is
  // never stored; sw_out = dgu [M, 2 ff] from sw_in = gu [M, 2 ff].
  const float* sc_m;  // fp8 operand form: per-row scales of A [M] and B [N] (applied to the accumulators)
  const float* sc_n;
  // fp8: optional per-(row, 128-k block) exponents of B (4-bit e, scale 2^-e relative to sc_n[row]; two rows per byte):
  // image [N / 128 groups][sc_e_group bytes], a group = [K / 128][64 B]; sc_e_group = round_up(K / 128 * 64, 4096)
  const uint8_t* sc_e;
  const int* sc_e_flag;  // device word, non-zero when any exponent of the image is non-zero (written by the quantiser); may be null
  int sc_e_group;
  int sw_mode, sw_ff;
  int gm;  // raster: tiles are visited in groups of gm tile-rows x all tile-columns (tile_of); 0 = automatic (4, or every tile row when there are <= 8)
  void* sw_out;
  const void* sw_in;
  int64_t sw_ldo, sw_ldi;
  // fp8 SwiGLU-backward dgrad (4-wave fp8 kernel): when non-null, the store phase also takes the row / column maxima of |dgu| as STORED (16-bit
  // rounded) by order-independent atomicMax on the values' bit patterns - amax_r [M], amax_c [2 ff], zeroed by the caller: what absmax_rc_k would read back
  unsigned* amax_r;
  unsigned* amax_c;
};

constexpr int BK = 64;

// tile index v of `nwg` -> (tm, tn).  Consecutive v go to consecutive XCDs (block b runs on XCD b % 8), so the tiles are
// re-numbered to give each XCD one contiguous run, visited in groups of 8 tile-rows (operand panels shared in that XCD's L2).
__device__ __forceinline__ void tile_of(const GemmArgs& g, int v, int nwg, int& tm, int& tn) {
  const int q = nwg >> 3, r = nwg & 7, xcd = v & 7, idx = v >> 3;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int GM = g.gm > 0 ? g.gm : (g.tiles_m <= 8 ? g.tiles_m : 4);  // tile rows per raster group (launcher: g_gemm_gm; 0 = automatic)
  const int per_group = GM * g.tiles_n;
  const int group = tile / per_group;
  const int first_m = group * GM;
  const int gsize = min(g.tiles_m - first_m, GM);
  const int in_g = tile - group * per_group;
  tm = first_m + in_g % gsize;
  tn = in_g / gsize;
}
// block id -> (tm, tn): one tile per block
__device__ __forceinline__ void tile_of_block(const GemmArgs& g, int& tm, int& tn) { tile_of(g, blockIdx.x, gridDim.x, tm, tn); }

// epilogue for 4 consecutive n of one row m (v = fp32 accumulators)
template <int DT>
__device__ __forceinline__ void epi_store4(const GemmArgs& g, int m, int n, float v0, float v1, float v2, float v3) {
  if (m >= g.M || n >= g.N) return;
  const int epi = g.epi;
  float v[4] = {v0, v1, v2, v3};
  if (g.vec_ok) {
    if (epi & MH_EPI_BIAS) {
      const uint2 bb = *(const uint2*)(g.bias + n);
      float b0, b1, b2, b3;
      unpack2<DT>(bb.x, b0, b1);
      unpack2<DT>(bb.y, b2, b3);
      v[0] += b0; v[1] += b1; v[2] += b2; v[3] += b3;
    }
    if (epi & MH_EPI_QUICK_GELU) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + __expf(-1.702f * v[r]));
    }
    if (epi & MH_EPI_RESIDUAL) {
      const uint2 rr = *(const uint2*)(g.resid + (int64_t)m * g.ldr + n);
      float r0, r1, r2, r3;
      unpack2<DT>(rr.x, r0, r1);
      unpack2<DT>(rr.y, r2, r3);
      v[0] += r0; v[1] += r1; v[2] += r2; v[3] += r3;
    }
    if (epi & MH_EPI_OUT_F32) {
      float4* dst = (float4*)((float*)g.C + (int64_t)m * g.ldc + n);
      if (epi & MH_EPI_ACCUM) {
        const float4 o = *dst;
        v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
      }
      *dst = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      uint2* dst = (uint2*)((uint16_t*)g.C + (int64_t)m * g.ldc + n);
      if (epi & MH_EPI_ACCUM) {
        const uint2 o = *dst;
        float o0, o1, o2, o3;
        unpack2<DT>(o.x, o0, o1);
        unpack2<DT>(o.y, o2, o3);
        v[0] += o0; v[1] += o1; v[2] += o2; v[3] += o3;
      }
      *dst = make_uint2(pack2<DT>(v[0], v[1]), pack2<DT>(v[2], v[3]));
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (n + r >= g.N) continue;
      float x = v[r];
      if (epi & MH_EPI_BIAS) x += ld16<DT>(g.bias[n + r]);
      if (epi & MH_EPI_QUICK_GELU) x = x / (1.0f + __expf(-1.702f * x));
      if (epi & MH_EPI_RESIDUAL) x += ld16<DT>(g.resid[(int64_t)m * g.ldr + n + r]);
      if (epi & MH_EPI_OUT_F32) {
        float* dst = (float*)g.C + (int64_t)m * g.ldc + n + r;
        if (epi & MH_EPI_ACCUM) x += *dst;
        *dst = x;
      } else {
        uint16_t* dst = (uint16_t*)g.C + (int64_t)m * g.ldc + n + r;
        if (epi & MH_EPI_ACCUM) x += ld16<DT>(*dst);
        *dst = (uint16_t)st16<DT>(x);
      }
    }
  }
}

// Compile-time epilogue: the flag tests of epi_store4 cost ~100 instructions per call when `epi` is a run-time
// value, i.e. ~50 k cycles for the 64 calls of a 128x128 wave quadrant - as much as 25 K-steps of MFMA work.  The
// kernels therefore branch ONCE on g.epi (epi_dispatch) into a store loop specialised for the common flag sets
// (vectorisable layouts only); everything else goes through the generic epi_store4.
template <int DT, int EPI>
struct EpiStoreFast {
  const GemmArgs& g;
  __device__ __forceinline__ void operator()(int m, int n, float v0, float v1, float v2, float v3) const {
    if (m >= g.M || n >= g.N) return;
    float v[4] = {v0, v1, v2, v3};
    if constexpr (EPI & MH_EPI_BIAS) {
      const uint2 bb = *(const uint2*)(g.bias + n);
      float b0, b1, b2, b3;
      unpack2<DT>(bb.x, b0, b1);
      unpack2<DT>(bb.y, b2, b3);
      v[0] += b0; v[1] += b1; v[2] += b2; v[3] += b3;
    }
    if constexpr (EPI & MH_EPI_QUICK_GELU) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + __expf(-1.702f * v[r]));
    }
    if constexpr (EPI & MH_EPI_RESIDUAL) {
      const uint2 rr = *(const uint2*)(g.resid + (int64_t)m * g.ldr + n);
      float r0, r1, r2, r3;
      unpack2<DT>(rr.x, r0, r1);
      unpack2<DT>(rr.y, r2, r3);
      v[0] += r0; v[1] += r1; v[2] += r2; v[3] += r3;
    }
    if constexpr (EPI & MH_EPI_OUT_F32) {
      float4* dst = (float4*)((float*)g.C + (int64_t)m * g.ldc + n);
      if constexpr (EPI & MH_EPI_ACCUM) {
        const float4 o = *dst;
        v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
      }
      *dst = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      uint2* dst = (uint2*)((uint16_t*)g.C + (int64_t)m * g.ldc + n);
      if constexpr (EPI & MH_EPI_ACCUM) {
        const uint2 o = *dst;
        float o0, o1, o2, o3;
        unpack2<DT>(o.x, o0, o1);
        unpack2<DT>(o.y, o2, o3);
        v[0] += o0; v[1] += o1; v[2] += o2; v[3] += o3;
      }
      *dst = make_uint2(pack2<DT>(v[0], v[1]), pack2<DT>(v[2], v[3]));
    }
  }
};
// The value transform alone (bias / quick-GELU / residual in fp32), for epilogues that stage the 16-bit result
// through LDS to get full-line global stores.  Requires the vectorisable layout (g.vec_ok) and m < M, n + 3 < N.
template <int DT, int EPI>
__device__ __forceinline__ void epi_xform4(const GemmArgs& g, int m, int n, float (&v)[4]) {
  if constexpr (EPI & MH_EPI_BIAS) {
    const uint2 bb = *(const uint2*)(g.bias + n);
    float b0, b1, b2, b3;
    unpack2<DT>(bb.x, b0, b1);
    unpack2<DT>(bb.y, b2, b3);
    v[0] += b0; v[1] += b1; v[2] += b2; v[3] += b3;
  }
  if constexpr (EPI & MH_EPI_QUICK_GELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + __expf(-1.702f * v[r]));
  }
  if constexpr (EPI & MH_EPI_RESIDUAL) {
    const uint2 rr = *(const uint2*)(g.resid + (int64_t)m * g.ldr + n);
    float r0, r1, r2, r3;
    unpack2<DT>(rr.x, r0, r1);
    unpack2<DT>(rr.y, r2, r3);
    v[0] += r0; v[1] += r1; v[2] += r2; v[3] += r3;
  }
}
// true when the 16-bit result can be written as 16-byte row segments (staged epilogues)
__device__ __forceinline__ bool epi_can_stage(const GemmArgs& g) {
  const int e = g.epi;
  const bool known = e == 0 || e == MH_EPI_RESIDUAL || e == MH_EPI_BIAS || e == (MH_EPI_BIAS | MH_EPI_QUICK_GELU) ||
                     e == (MH_EPI_BIAS | MH_EPI_RESIDUAL);
  return known && g.vec_ok && (g.N % 8 == 0) && (g.ldc % 8 == 0) && ((((uintptr_t)g.C) & 15u) == 0);
}

// (inlined: an out-of-line call would give the kernels a scratch segment, which costs far more at wave launch)
template <int DT>
struct EpiStoreGeneric {
  const GemmArgs& g;
  __device__ __forceinline__ void operator()(int m, int n, float v0, float v1, float v2, float v3) const {
    epi_store4<DT>(g, m, n, v0, v1, v2, v3);
  }
};
// body(store): the kernel's loop over its accumulator tiles, calling store(m, n, v0, v1, v2, v3)
template <int DT, typename F>
__device__ __forceinline__ void epi_dispatch(const GemmArgs& g, F&& body) {
  if (g.vec_ok) {
    switch (g.epi) {
      case 0: body(EpiStoreFast<DT, 0>{g}); return;
      case MH_EPI_RESIDUAL: body(EpiStoreFast<DT, MH_EPI_RESIDUAL>{g}); return;
      case MH_EPI_BIAS: body(EpiStoreFast<DT, MH_EPI_BIAS>{g}); return;
      case MH_EPI_BIAS | MH_EPI_QUICK_GELU: body(EpiStoreFast<DT, MH_EPI_BIAS | MH_EPI_QUICK_GELU>{g}); return;
      case MH_EPI_BIAS | MH_EPI_RESIDUAL: body(EpiStoreFast<DT, MH_EPI_BIAS | MH_EPI_RESIDUAL>{g}); return;
      case MH_EPI_OUT_F32: body(EpiStoreFast<DT, MH_EPI_OUT_F32>{g}); return;
      case MH_EPI_OUT_F32 | MH_EPI_ACCUM: body(EpiStoreFast<DT, MH_EPI_OUT_F32 | MH_EPI_ACCUM>{g}); return;
      case MH_EPI_BIAS | MH_EPI_OUT_F32 | MH_EPI_ACCUM: body(EpiStoreFast<DT, MH_EPI_BIAS | MH_EPI_OUT_F32 | MH_EPI_ACCUM>{g}); return;
      case MH_EPI_ACCUM: body(EpiStoreFast<DT, MH_EPI_ACCUM>{g}); return;
      default: break;
    }
  }
  body(EpiStoreGeneric<DT>{g});
}

// defined in gemm256.hip
int launch_gemm_nt_256(const GemmArgs& g, int dt, hipStream_t stream);
// operand layouts: K-contiguous (0) or K-strided (1), see gemm256.hip
int launch_gemm_256(const GemmArgs& g, int dt, int a_kstrided, int b_kstrided, hipStream_t stream);
int launch_gemm_nt_w4(const GemmArgs& g, int dt, hipStream_t stream, int var = 0);      // tools/dev_arms/gemm256w4.hip (first 4-wave arm)
// gemm_w4.hip: 4 waves x 128x128 per wave, all operand layouts; staged 16-bit epilogue (plain / residual / accumulate), fp32 stores, and the
// fused RoPE / SwiGLU / SwiGLU-backward forms on the fp32 accumulators; split-K (fp32 partials); K tails of K-strided operands
bool w4_can_run(const GemmArgs& g, int a_kstrided, int b_kstrided);
bool w4_has_kernel(const GemmArgs& g, int a_kstrided, int b_kstrided);  // the (epilogue kind, layout) pair is instantiated
int launch_gemm_w4(const GemmArgs& g, int dt, int a_kstrided, int b_kstrided, hipStream_t stream, int half = 0);  // half: 128-row block tiles
bool w4_has_half(const GemmArgs& g, int a_kstrided, int b_kstrided);
int launch_gemm_w4_grouped(const GemmArgs* probs, int n, int dt, int gm, hipStream_t stream);  // n weight gradients (TN) over one K, one launch
bool w4_f8_can_run(const GemmArgs& g);  // fp8 operands (no block exponents), see gemm_w4.hip
bool w4_f8_is_fused(const GemmArgs& g);  // RoPE / SwiGLU / fp32-store kinds (own policy bit)
int launch_gemm_w4_f8(const GemmArgs& g, int dt, hipStream_t stream);
int launch_absmax_rc(const void* x, int64_t ldx, unsigned* rmax, unsigned* cmax, int R, int C, int dt, hipStream_t stream);  // fp8_quant.hip: row and column maxima of |x| in one read
int launch_gemm_nt_256_f8(const GemmArgs& g, int dt, hipStream_t stream);  // gemm256.hip, fp8 operands + f8f6f4 MFMA
int launch_gemm_nt_w8(const GemmArgs& g, int dt, hipStream_t stream);      // gemm256w8.hip (8 waves, dense asm stream)
int launch_gemm_nt_256_m32(const GemmArgs& g, int dt, hipStream_t stream);  // gemm256_m32.hip (32x32x16 arm)
extern int g_m32_kcut;

}  // namespace mhgemm
