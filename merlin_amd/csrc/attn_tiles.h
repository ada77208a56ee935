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

// (one entry of the table above for a run-time d-block i: a table indexed by a run-time value would live in scratch memory)
template <int D>
__device__ __forceinline__ unsigned tr_frag_offset_one(int lane, int i, int rr) {
  const int G = lane >> 4, hi = G >> 1, db = G & 1;
  const int y = (lane & 15) >> 2, x = (lane & 3) >> 1, z = lane & 1;
  const int row = 8 * rr + 4 * hi + y;
  const int chunk = (4 * i + 2 * db + x) ^ TileSwz<D>::f(row);
  return (unsigned)((4 * hi + y) * (D * 2) + (chunk << 4) + 8 * z);
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
// (round 3 form, kept for the opt-in ping-pong forward arm attn_fwd3.hip; the production kernels use stage_rows_buf below)
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


// ---- the same tile copy through a buffer descriptor: NO vector arithmetic per tile --------------------------------------------------------
// The copies above keep one 64-bit pointer per thread and copy (or re-derive it, clamped, with quarter-rate multiplies) and advance all of
// them every tile - 16-24 64-bit VALU operations per K|V tile in a loop whose VALU stream is as long as its MFMA stream.  Here the tile's
// position is entirely scalar: a descriptor whose BASE is the tile's first row (64-bit SALU add).  The per-thread part is ONE register:
// thread t of a copy instruction takes 16-byte chunk (t % CPR) ^ swz of row t / CPR, and instruction i of a tile adds i * RPL rows - a scalar
// (soffset), because the swizzle only sees row % 16 and RPL is a multiple of 16.
// Rows past the end of the batch element (the last tile of a sequence whose length is not a multiple of the tile) must not be READ - beyond
// the last batch element lies memory that is not ours - and arrive as ZEROS (every kernel masks them; the clamped copies of the last row the
// forms above deliver needed the same masks).  The hardware checks base + voffset against num_records but NOT soffset, so that tile takes one
// descriptor per copy instruction: base at the instruction's first row, num_records = what is left of the batch element from there.
// M0 is saved and restored inside the statement (hipcc owns it: the backward kernels' global_load_lds builtins set it too); `s_nop 4` opens
// the statement for the case that hipcc produced one of the scalar operands with v_readfirstlane (cdna guide, inline asm item 2).
typedef int i32x4_t __attribute__((ext_vector_type(4)));
template <int D>
struct RowSrc {
  uint64_t base;       // byte address of (row 0, col 0) of this (batch, head)
  unsigned row_bytes;  // row stride in bytes
  unsigned span;       // bytes from base to the end of the batch element's last row = rows * row_bytes (< 2^31, checked by the launchers)
  int rows;            // rows of the batch element (S)
  unsigned voff;       // this thread's byte offset inside a copy instruction's row group
  unsigned step[3];    // soffset of copy instructions 1..3 of a tile
};
template <int D>
__device__ __forceinline__ RowSrc<D> row_src(const uint16_t* gbase, int64_t ld, int S, int tid) {
  constexpr int CPR = D / 8, RPL = 256 / CPR;
  RowSrc<D> r;
  r.base = (uint64_t)(uintptr_t)gbase;
  r.row_bytes = (unsigned)ld * 2u;
  r.rows = S;
  r.span = (unsigned)S * r.row_bytes;
  const int row = tid / CPR, cc = tid % CPR;
  r.voff = (unsigned)row * r.row_bytes + (unsigned)((cc ^ TileSwz<D>::f(row)) * 16);
#pragma unroll
  for (int i = 0; i < 3; ++i) r.step[i] = (unsigned)((i + 1) * RPL) * r.row_bytes;
  return r;
}
__device__ __forceinline__ i32x4_t row_srd(uint64_t base, unsigned nrec) {
  return i32x4_t{(int)(uint32_t)base, (int)(uint32_t)((base >> 32) & 0xffffu), (int)nrec, 0x00020000};
}
// lds_wave = LDS byte address of the tile + wave * 1024 (wave-uniform); r0 = first row of the tile (wave-uniform, < rows)
template <int D, int ROWS>
__device__ __forceinline__ void stage_rows_buf(const RowSrc<D>& src, int r0, unsigned lds_wave) {
  constexpr int NLD = ROWS * (D / 8) / 256;
  static_assert(NLD == 2 || NLD == 4, "64-row tiles of D = 64 / 128");
  const unsigned adv = (unsigned)r0 * src.row_bytes;
  unsigned keep;
  if (r0 + ROWS <= src.rows) {  // wave-uniform; every row of the tile exists: one descriptor, nothing to clamp
    const i32x4_t rs = row_srd(src.base + adv, src.span - adv);
    if constexpr (NLD == 4) {
      asm volatile(
          "s_nop 4\n\ts_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
          "s_add_u32 m0, %3, 0x1000\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
          "s_add_u32 m0, %3, 0x2000\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %5 offen lds\n\t"
          "s_add_u32 m0, %3, 0x3000\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %6 offen lds\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(src.voff), "s"(rs), "s"(lds_wave), "s"(src.step[0]), "s"(src.step[1]), "s"(src.step[2])
          : "memory", "scc");
    } else {
      asm volatile(
          "s_nop 4\n\ts_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
          "s_add_u32 m0, %3, 0x1000\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(src.voff), "s"(rs), "s"(lds_wave), "s"(src.step[0])
          : "memory", "scc");
    }
  } else {  // the sequence's last, partial tile: one descriptor per copy instruction (range-checked row by row)
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const unsigned adv_i = adv + (i ? src.step[i - 1] : 0u);
      const long long left = (long long)src.span - (long long)adv_i;
      const i32x4_t rs = row_srd(src.base + adv_i, left > 0 ? (unsigned)left : 0u);
      const unsigned dst = lds_wave + (unsigned)i * 0x1000u;
      asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(src.voff), "s"(rs), "s"(dst)
                   : "memory");
    }
  }
}

// ---- row-per-lane epilogue stores, widened (cdna guide T21) --------------------------------------------------------------------------
// After a 32x32 MFMA chain a lane (row l31, half hi) holds, per accumulator block i and group g, the 4 channels 32 i + 8 g + 4 hi .. + 3 of
// its row: the natural store is DBLK * 4 stores of 8 bytes.  One v_permlane32_swap per dword exchanges the upper half-wave's group g with
// the lower half-wave's group g + 1, after which the lower lane holds channels 8 g .. 8 g + 7 and the upper lane 8 (g + 1) .. 8 (g + 1) + 7
// of the row: DBLK * 2 stores of 16 bytes, same bytes, same addresses, half the store instructions (the epilogues are store-ISSUE
// bound: MI355X_MICROARCH "attention epilogue store tail").  `rowp` = this lane's row (16-byte aligned, checked by the launchers through
// `wide`); v(i, r) = the fp32 value of accumulator block i, register r, after whatever the caller applies; !valid rows are written as zeros.
template <int DT, int DBLK, typename F>
__device__ __forceinline__ void store_row_wide(uint16_t* rowp, int hi, bool valid, F&& v) {
#pragma unroll
  for (int i = 0; i < DBLK; ++i)
#pragma unroll
    for (int gp = 0; gp < 4; gp += 2) {
      uint32_t ax = 0, ay = 0, bx = 0, by = 0;
      if (valid) {
        ax = pack2<DT>(v(i, 4 * gp + 0), v(i, 4 * gp + 1)); ay = pack2<DT>(v(i, 4 * gp + 2), v(i, 4 * gp + 3));
        bx = pack2<DT>(v(i, 4 * gp + 4), v(i, 4 * gp + 5)); by = pack2<DT>(v(i, 4 * gp + 6), v(i, 4 * gp + 7));
      }
      const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);  // (validity is a property of the ROW: both half-waves agree)
      const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
      *(uint4*)(rowp + 32 * i + 8 * gp + 8 * hi) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
    }
}
template <int DT, int DBLK, typename F>
__device__ __forceinline__ void store_row_narrow(uint16_t* rowp, int hi, bool valid, F&& v) {
#pragma unroll
  for (int i = 0; i < DBLK; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint2 w = make_uint2(0, 0);
      if (valid) w = make_uint2(pack2<DT>(v(i, 4 * g + 0), v(i, 4 * g + 1)), pack2<DT>(v(i, 4 * g + 2), v(i, 4 * g + 3)));
      *(uint2*)(rowp + 32 * i + 8 * g + 4 * hi) = w;
    }
}

}  // namespace mhattn
