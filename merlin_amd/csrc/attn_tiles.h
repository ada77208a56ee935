// Shared LDS-tile machinery of the attention kernels (gfx950).
//
// Every operand tile is staged ROW-MAJOR, [rows][D] 16-bit (rows = keys or queries), by global_load_lds, and
// is consumed in one of two ways by MFMA 32x32x16:
//   * row fragments  (MFMA row index = tile row, contraction over d): one ds_read_b128 per fragment;
//   * TRANSPOSED fragments (MFMA row index = d, contraction over the tile rows): two ds_read_b64_tr_b16
//     transpose-reads per fragment - this is what replaces the pre-transposed V^T / Q^T / dO^T / K^T copies.
//     The contraction ("slot") order is the one implied by feeding a 32x32 score accumulator straight back as
//     the B operand: slot (hi, e) of k-step s  <->  tile row 16*s + 4*hi + e (e < 4), 16*s + 8 + 4*hi + (e-4).
//
// Bank swizzle (XOR on the 16-byte chunk index, applied to the global source address of the lane-linear
// global_load_lds image and again on every read):
//   D = 128 (256-B rows, 16 chunks):  chunk ^= ((row & 3) << 2) | ((row >> 2) & 3)
//   D =  64 (128-B rows,  8 chunks):  v = (row >> 1) & 7;  chunk ^= ((v & 1) << 2) | (v >> 1)
// Both are bijections of the row index within 16 rows (row fragments: a 16-lane ds_read_b128 group hits 16
// distinct 16-B slots) and send 4 consecutive rows to 4 different 64-B windows (transpose-reads: the 32 lanes
// of a half-wave hit 16 distinct slots).  Conflict-free for both access kinds.
#pragma once
#include <type_traits>

#include "mh_common.h"

#ifndef MH_ATTN_SETPRIO
#define MH_ATTN_SETPRIO 1  // raise the wave's priority over its MFMA clusters in the two-waves-per-SIMD kernels (A/B: -DMH_ATTN_SETPRIO=0)
#endif

namespace mhattn {

// cdna guide T5: in the kernels that run two unsynchronised waves per SIMD (forward, dQ) the wave inside an MFMA cluster should win
// issue arbitration over its partner's softmax / address arithmetic, so the matrix pipe stays fed
__device__ __forceinline__ void prio_mfma(bool on) {
#if MH_ATTN_SETPRIO
  if (on) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
}

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

template <int D> struct TileSwz;
template <> struct TileSwz<128> {
  static __device__ __forceinline__ int f(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
};
template <> struct TileSwz<64> {
  static __device__ __forceinline__ int f(int row) { const int v = (row >> 1) & 7; return ((v & 1) << 2) | (v >> 1); }
};

template <int OFF>
__device__ __forceinline__ void lds_read64_tr(u32x2_t& d, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}

// Per-lane byte offsets (relative to a tile base) for ROW fragments of a [rows][D] tile: fragment row = l31 (+32*blk via
// immediate), k-step ks -> chunk 2*ks + hi.
template <int D>
__device__ __forceinline__ void row_frag_offsets(int l31, int hi, unsigned (&off)[D / 16]) {
  const int sw = TileSwz<D>::f(l31);
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) off[ks] = l31 * (D * 2) + (((2 * ks + hi) ^ sw) << 4);
}

// Per-lane byte offsets for TRANSPOSED fragments: index v = i (d-block, 0..D/32-1) * 2 + rr (which half of the 8 slots).
// A fragment (i, s) is  {tr(off[2i+0] + s*16 rows), tr(off[2i+1] + s*16 rows + 8 rows)}; the s / rr row steps are
// immediates (16*s + 8*rr rows), everything lane-dependent (incl. the XORed chunk) is in off[].
template <int D>
__device__ __forceinline__ void tr_frag_offsets(int lane, unsigned (&off)[D / 16]) {
  const int G = lane >> 4, hi = G >> 1, db = G & 1;
  const int y = (lane & 15) >> 2, x = (lane & 3) >> 1, z = lane & 1;
  constexpr int RB = D * 2;
#pragma unroll
  for (int i = 0; i < D / 32; ++i)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = 8 * rr + 4 * hi + y;  // (+16*s via immediate; the swizzle only sees row & 15)
      const int chunk = (4 * i + 2 * db + x) ^ TileSwz<D>::f(row);
      off[2 * i + rr] = (4 * hi + y) * RB + (chunk << 4) + 8 * z;  // row part without rr/s (added as immediates)
    }
}

// global_load_lds of one [ROWS][D] row-major tile: gbase points at (row 0, col 0) of the tile for this (b,h);
// row r of the tile is at gbase + min(r0 + r, rmax) * ld.  256 threads, wave-uniform LDS destination.
// Tiles that lie wholly inside the sequence (all but the last one) need no clamp: their addresses are a wave-uniform tile base
// (scalar arithmetic) plus per-thread byte offsets that do not depend on the tile - stage_offsets() computes those once per
// kernel.  Computed per tile, the clamped form cost 16 v_mul_lo_u32 + 8 v_mad_u64_u32 (quarter-rate) and ~45 more integer VALU
// instructions per K|V tile, a quarter of the forward kernel's VALU work.
template <int D, int ROWS>
struct StageOffs {
  unsigned off[ROWS * (D / 8) / 256];
};
template <int D, int ROWS>
__device__ __forceinline__ StageOffs<D, ROWS> stage_offsets(int64_t ld, int tid) {
  constexpr int CPR = D / 8;
  constexpr int NLD = ROWS * CPR / 256;
  StageOffs<D, ROWS> o;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int qd = i * 256 + tid;
    const int row = qd / CPR, cc = qd % CPR;
    const int c = cc ^ TileSwz<D>::f(row);
    o.off[i] = ((unsigned)row * (unsigned)ld + (unsigned)(c * 8)) * 2u;  // bytes; < 2^31 for any row stride this library accepts
  }
  return o;
}
template <int D, int ROWS>
__device__ __forceinline__ void stage_rows(const uint16_t* gbase, int64_t ld, int r0, int rmax, char* lds_tile, int tid, int wave) {
  constexpr int CPR = D / 8;
  constexpr int NLD = ROWS * CPR / 256;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int qd = i * 256 + tid;
    const int row = qd / CPR, cc = qd % CPR;
    const int c = cc ^ TileSwz<D>::f(row);
    glds16(gbase + (int64_t)min(r0 + row, rmax) * ld + c * 8, lds_tile + (i * 256 + wave * 64) * 16);
  }
}
// (the kernels with registers to spare - not the dQ and the stand-alone dK kernel, at 251 / 256 VGPRs - use this form)
template <int D, int ROWS>
__device__ __forceinline__ void stage_rows(const uint16_t* gbase, int64_t ld, int r0, int rmax, char* lds_tile, int tid, int wave,
                                           const StageOffs<D, ROWS>& so) {
  constexpr int NLD = ROWS * (D / 8) / 256;
#ifndef MH_STAGE_CLAMPED_ONLY  // (development A/B: build with -DMH_STAGE_CLAMPED_ONLY for the per-tile address arithmetic everywhere)
  if (r0 + ROWS - 1 <= rmax) {  // wave-uniform
    const char* tb = (const char*)(gbase + (int64_t)r0 * ld);
#pragma unroll
    for (int i = 0; i < NLD; ++i) glds16(tb + so.off[i], lds_tile + (i * 256 + wave * 64) * 16);
    return;
  }
#endif
  stage_rows<D, ROWS>(gbase, ld, r0, rmax, lds_tile, tid, wave);
}

}  // namespace mhattn
