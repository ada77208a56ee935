// gemm_nt_256: 256x256x64-tile bf16/f16 MFMA GEMM with a counted-vmcnt, two-group staggered pipeline
// (gfx950).  See gemm.hip for what is common to both GEMM kernels (swizzle, swapped operands, epilogue).
//
// Geometry: 512 threads = 8 waves (wm = wave>>2 in {0,1}, wn = wave&3); a K-tile is four 16-KiB
// half-tiles in one of two LDS stages (128 KiB): A0/A1 = A rows 0-127 / 128-255, B0/B1 likewise.  A wave
// owns rows {wm*64..+64} of EACH A half and columns {wn*32..+32} of EACH B half: its 128x64 output is four
// 64x32 quadrants visited per K-tile as (A0,B0) (A0,B1) (A1,B1) (A1,B0) = four phases of 16 MFMA 16x16x32,
// with LDS fragment reads A0+B0 | B1 | A1 | none (B0 and the current A fragments stay in registers).
//
// Pipeline.  Every phase is two segments separated by raw s_barriers:
//     A-seg: ds_read the phase's fragments; issue ONE half-tile of global_load_lds (2 per thread);
//            s_waitcnt vmcnt(8)  (own loads of the half-tile the NEXT phase reads have landed; the four
//            newest half-tiles stay in flight - never vmcnt(0) in the loop)
//     B-seg: s_waitcnt lgkmcnt(0); 16 MFMAs
//   The wm=1 waves run one barrier behind the wm=0 waves (one wave of each group per SIMD), so on every
//   SIMD one wave's MFMA segment overlaps the other's LDS-read/issue segment.
//   Load order = consumption order A0 B0 B1 A1; a half-tile's LDS slot is refilled two phases after its
//   last read (so both groups are done with it): phase 0 issues B1(t+1), 1: A1(t+1), 2: A0(t+2), 3: B0(t+2),
//   i.e. every load has >= 5 phases (> 1 K-tile of MFMA time) to land.
//   RAW: a half-tile is read one phase after every thread's counted wait for it, with a barrier between.
//   WAR: a slot is refilled >= 2 phases after its last read; reads complete (lgkmcnt(0)) in the B-seg of
//   their own phase, and the lagging group's B-seg of phase q ends before the leading group's A-seg of q+2.
//
// hipcc specifics: the fragment reads are inline-asm `ds_read_b128` (hipcc would otherwise put
// `s_waitcnt vmcnt(0)` in front of every ds_read that follows a global_load_lds, draining the pipeline),
// so their completion is waited for by hand (lgkmcnt(0) + sched_barrier, cdna guide rule 18).
//
// Operand layouts.  Each operand is either K-contiguous (rows = m or n, the forward "NT" case) or K-STRIDED
// (memory is [K][M] / [K][N], the non-contracted index contiguous): dgrad dX = dY * W reads W [N_out][K_in] as
// a K-strided B, wgrad dW = dY^T * X reads both operands K-strided - no transposed copies are ever made.
// A K-strided half-tile is staged as [64 k][128 m] (256-B rows) and its MFMA fragments are fetched with the
// gfx950 transpose read: two `ds_read_b64_tr_b16` give a lane 8 k-values of its own m (lane i of a 16-lane group
// supplies the address of row i>>2, columns 4*(i&3)..+3 of a 4x16 block and receives column i).  Swizzle: the
// 32-byte column chunk is XORed with f(k) = (k&3) | ((k>>3)&1)<<2, which is a per-lane constant for the tr
// read pattern and makes both 32-lane halves of every ds_read_b64_tr_b16 hit 8 distinct 32-B slots.
#include <type_traits>

#include "gemm_common.h"

namespace mhgemm {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int DT>
__device__ __forceinline__ f32x4_t mfma16v(u32x4 a, u32x4 b, f32x4_t c) {
  if constexpr (DT == MH_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

constexpr int HALF_BYTES = 128 * BK * 2;  // 16 KiB: 128 rows x 64 k
constexpr int STAGE256 = 4 * HALF_BYTES;  // A0 A1 B0 B1
constexpr int H_A0 = 0, H_A1 = 1, H_B0 = 2, H_B1 = 3;
constexpr int C_ROW = 528;                 // C staging row: 256 x 16-bit + 16 B pad
constexpr int LDS256 = 256 * C_ROW;        // >= 2 * STAGE256: two stages during the loop, the C tile after it
constexpr int F8_EXP_BYTES = 2 * 12288;    // fp8: block-exponent images of the two B groups, K <= 24 576 (160 KiB - LDS256 = 28 672)

#define MH_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define MH_BAR()                         \
  do {                                   \
    __builtin_amdgcn_sched_barrier(0);   \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)
#define MH_LGKM0()                                        \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);                    \
  } while (0)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ void lds_read64_tr(u32x2& d, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
// fragment i (0..NFRAG-1) x k-step ks (0..1) of a K-strided half-tile at byte offset BASE: rows k = 32*ks + 8*g +
// (i16>>2) (+4 for the second read), per-fragment lane address at[i]
template <int BASE, int NFRAG, typename FR>
__device__ __forceinline__ void read_frags_tr(FR& fr, const unsigned* at) {
#pragma unroll
  for (int i = 0; i < NFRAG; ++i) {
    u32x2 a0, a1, b0, b1;
    lds_read64_tr<BASE + 0>(a0, at[i]);
    lds_read64_tr<BASE + 1024>(a1, at[i]);
    lds_read64_tr<BASE + 8192>(b0, at[i]);
    lds_read64_tr<BASE + 8192 + 1024>(b1, at[i]);
    fr[i][0] = u32x4{a0[0], a0[1], a1[0], a1[1]};
    fr[i][1] = u32x4{b0[0], b0[1], b1[0], b1[1]};
  }
}
#define DSR(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
// 4 row-fragments (i*2048 apart) of one half-tile at byte offset BASE, k-step addresses a0/a1
#define READ_A(BASE)                                                                        \
  do {                                                                                      \
    DSR(af[0][0], aA0, BASE + 0);    DSR(af[0][1], aA1, BASE + 0);                          \
    DSR(af[1][0], aA0, BASE + 2048); DSR(af[1][1], aA1, BASE + 2048);                       \
    DSR(af[2][0], aA0, BASE + 4096); DSR(af[2][1], aA1, BASE + 4096);                       \
    DSR(af[3][0], aA0, BASE + 6144); DSR(af[3][1], aA1, BASE + 6144);                       \
  } while (0)
#define READ_B(bf, BASE)                                                                    \
  do {                                                                                      \
    DSR(bf[0][0], aB0, BASE + 0);    DSR(bf[0][1], aB1, BASE + 0);                          \
    DSR(bf[1][0], aB0, BASE + 2048); DSR(bf[1][1], aB1, BASE + 2048);                       \
  } while (0)
// fp8 operands (F8): a 128-byte LDS row holds 128 k instead of 64, and the two 16-byte pieces a lane reads per row
// (chunks kq and 4 + kq) are exactly the 32 bytes v_mfma_scale_f32_16x16x128_f8f6f4 wants from lane group kq (layout and
// rate probed in tools/probes/f8f6f4_probe.py): one MFMA per accumulator tile and K-tile, same bytes per phase, twice
// the k.  Hardware block scales are left at 1.0; per-row scales are applied to the accumulators after the loop.
typedef int i32x8_t __attribute__((ext_vector_type(8)));
// `sa` = E8M0 block scale (2^(sa-127)) of the FIRST operand's 128 k, one value per lane = per row (lane & 15) of the fragment; every
// lane group of a row supplies the same exponent, i.e. one scale per (row, 128-k block): the per-128-block weight scales of BASELINE
// cfg 5 applied by the MFMA itself (probe: profiles/r01_f8f6f4_probe.txt).  The second operand's block scale stays 1.0.
__device__ __forceinline__ f32x4_t mfma_f8(const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1, f32x4_t c, int sa) {
  const i32x8_t a = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
  const i32x8_t b = {(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, 127);
}
__device__ __forceinline__ void lds_read_u8(unsigned& d, unsigned addr) { asm volatile("ds_read_u8 %0, %1" : "=v"(d) : "v"(addr)); }
#ifdef MH_GEMM_MFMA32_SPEEDTEST
// SPEED PROBE ONLY (wrong results): the same fragments, registers and issue slots, but 8 32x32x16 MFMAs per quad instead of 16
// 16x16x32 - what the main loop would run at with the wider instruction (tools/probes/mfma_probe.py: a lone 16x16x32 stream tops out
// at 91 % of peak with two waves per SIMD, 32x32x16 at 99 %).
typedef float f32x16p_t __attribute__((ext_vector_type(16)));
#define MFMA_QUAD_16(MH, NH, bf)                                                            \
  do {                                                                                      \
    f32x16p_t q0_, q1_;                                                                     \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                         \
      q0_[e] = acc[MH][0][NH][0][e]; q0_[4 + e] = acc[MH][0][NH][1][e]; q0_[8 + e] = acc[MH][1][NH][0][e]; q0_[12 + e] = acc[MH][1][NH][1][e]; \
      q1_[e] = acc[MH][2][NH][0][e]; q1_[4 + e] = acc[MH][2][NH][1][e]; q1_[8 + e] = acc[MH][3][NH][0][e]; q1_[12 + e] = acc[MH][3][NH][1][e]; \
    }                                                                                       \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                        \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                       \
        q0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bf[j][ks]), __builtin_bit_cast(bf16x8_t, af[2 * j][ks]), q0_, 0, 0, 0);     \
        q1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bf[j][ks]), __builtin_bit_cast(bf16x8_t, af[2 * j + 1][ks]), q1_, 0, 0, 0); \
      }                                                                                     \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                         \
      acc[MH][0][NH][0][e] = q0_[e]; acc[MH][0][NH][1][e] = q0_[4 + e]; acc[MH][1][NH][0][e] = q0_[8 + e]; acc[MH][1][NH][1][e] = q0_[12 + e]; \
      acc[MH][2][NH][0][e] = q1_[e]; acc[MH][2][NH][1][e] = q1_[4 + e]; acc[MH][3][NH][0][e] = q1_[8 + e]; acc[MH][3][NH][1][e] = q1_[12 + e]; \
    }                                                                                       \
  } while (0)
#else
#define MFMA_QUAD_16(MH, NH, bf)                                                            \
  do {                                                                                      \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                        \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                         \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                       \
          acc[MH][i][NH][j] = mfma16v<DT>(bf[j][ks], af[i][ks], acc[MH][i][NH][j]);         \
  } while (0)
#endif
#define MFMA_QUAD(MH, NH, bf)                                                               \
  do {                                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                          \
    if constexpr (F8) {                                                                     \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                         \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                       \
          acc[MH][i][NH][j] = mfma_f8(bf[j][0], bf[j][1], af[i][0], af[i][1], acc[MH][i][NH][j], ESC ? esc[NH][j] : 127); \
    } else {                                                                                \
      MFMA_QUAD_16(MH, NH, bf);                                                             \
    }                                                                                       \
    __builtin_amdgcn_s_setprio(0);                                                          \
  } while (0)

template <int DT, bool AKS, bool BKS, bool F8 = false>
__global__ __launch_bounds__(512, 2) void gemm_nt_256(GemmArgs g) {  // (g by value: the split-K block offsets its own copy)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // PERSISTENT tile loop: block b computes tiles b, b + gridDim.x, ... (grid = one block per CU, or one per tile).  The
  // first K-tile of the next output tile is fetched while the finished one is still being written out (below).
  const int ntiles = g.tiles_m * g.tiles_n;
  // split-K: block y of grid.y takes K-tiles [kt0, kt0 + nk) and writes its own fp32 partial tile
  const int nk_total = (g.K + BK - 1) / BK;
  const int per_split = (nk_total + g.splits - 1) / g.splits;
  const int kt0 = blockIdx.y * per_split;
  const int nk = min(per_split, nk_total - kt0);
  if (g.splits > 1) g.C = (float*)g.C + (int64_t)blockIdx.y * g.c_split;
  // K % 64 != 0 (both operands K-strided only): rows k >= K of the last K-tile are read from a row of zeros
  const int k_tail = g.K - (nk_total - 1) * BK;  // valid rows of the global last K-tile (64 when K % 64 == 0)

  // staging sources of output tile v: half-tile h, instruction i: 16-byte chunk qd = i*512 + tid of the half-tile's 1024
  const uint16_t* src[4][2];
  int s_m0, s_n0, s_nB1;  // origin of the tile `src` points at
  auto setup = [&](int v) {
    int tm, tn;
    tile_of(g, v, ntiles, tm, tn);
    const int m0 = tm * 256, n0 = tn * (g.sw_mode == 1 ? 128 : 256);
    const int nB1 = g.sw_mode == 1 ? g.sw_ff + n0 : n0 + 128;  // first B row of the second half-tile
    s_m0 = m0; s_n0 = n0; s_nB1 = nB1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int qd = i * 512 + tid;
      if constexpr (!AKS) {
        const int row = qd >> 3, cc = qd & 7;
        const int c = (cc ^ ((row >> 1) & 7)) * 8;
        src[H_A0][i] = g.A + (int64_t)min(m0 + row, g.M - 1) * g.lda + c;
        src[H_A1][i] = g.A + (int64_t)min(m0 + 128 + row, g.M - 1) * g.lda + c;
      } else {  // [64 k][128 m]: row = k, 16 chunks per row, 32-byte chunk index swizzled by f(k)
        const int k = qd >> 4, cc = qd & 15;
        const int col = ((((cc >> 1) ^ ((k & 3) | (((k >> 3) & 1) << 2))) << 1) | (cc & 1)) * 8;
        src[H_A0][i] = g.A + (int64_t)k * g.lda + min(m0 + col, g.M - 8);
        src[H_A1][i] = g.A + (int64_t)k * g.lda + min(m0 + 128 + col, g.M - 8);
      }
      if constexpr (!BKS) {
        const int row = qd >> 3, cc = qd & 7;
        const int c = (cc ^ ((row >> 1) & 7)) * 8;
        src[H_B0][i] = g.B + (int64_t)min(n0 + row, g.N - 1) * g.ldb + c;
        src[H_B1][i] = g.B + (int64_t)min(nB1 + row, g.N - 1) * g.ldb + c;
      } else {
        const int k = qd >> 4, cc = qd & 15;
        const int col = ((((cc >> 1) ^ ((k & 3) | (((k >> 3) & 1) << 2))) << 1) | (cc & 1)) * 8;
        src[H_B0][i] = g.B + (int64_t)k * g.ldb + min(n0 + col, g.N - 8);
        src[H_B1][i] = g.B + (int64_t)k * g.ldb + min(n0 + 128 + col, g.N - 8);
      }
    }
  };
  const int64_t a_kstep = AKS ? (int64_t)BK * g.lda : (int64_t)BK;  // elements per K-tile
  const int64_t b_kstep = BKS ? (int64_t)BK * g.ldb : (int64_t)BK;
  const uint16_t* zsrc = g.zero_row + lane * 8;
  const bool ztail[2] = {(AKS && BKS) && ((tid >> 4) >= k_tail), (AKS && BKS) && (((512 + tid) >> 4) >= k_tail)};
  // issue half-tile h of K-tile kt (kt clamped to the last tile: uniform load count; the re-loads of the
  // tail only ever target slots nobody reads again)
  auto issue = [&](int h, int kt) {
    const int ktc = kt0 + min(kt, nk - 1);
    const int64_t koff = (int64_t)ktc * (h < 2 ? a_kstep : b_kstep);
    char* dst = smem + (kt & 1) * STAGE256 + h * HALF_BYTES + wave * 1024;
    if constexpr (AKS && BKS) {
      const bool last = (ktc == nk_total - 1);
      glds16((last && ztail[0]) ? zsrc : src[h][0] + koff, dst);
      glds16((last && ztail[1]) ? zsrc : src[h][1] + koff, dst + 8192);
    } else {
      glds16(src[h][0] + koff, dst);
      glds16(src[h][1] + koff, dst + 8192);
    }
  };

  f32x4_t acc[2][4][2][2];  // [mh][i][nh][j]

  const int frow = lane & 15;
  const int swz = (frow >> 1) & 7;
  const int kq = lane >> 4;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned a_row = lds0 + (wm * 64 + frow) * 128;  // K-contiguous: row inside an A half-tile
  const unsigned b_row = lds0 + (wn * 32 + frow) * 128;  // row inside a B half-tile
  const unsigned c0 = ((0 + kq) ^ swz) << 4, c1 = ((4 + kq) ^ swz) << 4;  // k-step 0 / 1 chunk offsets
  // K-strided (transpose reads): lane points at row k = 8*kq + (frow>>2), columns 4*(frow&3)..+3 of its fragment
  const unsigned t_row = lds0 + (kq * 8 + (frow >> 2)) * 256 + (frow & 3) * 8;
  const unsigned fx = (frow >> 2) | ((kq & 1) << 2);
  unsigned a_t[4], b_t[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_t[i] = t_row + (((wm * 4 + i) ^ fx) << 5);
#pragma unroll
  for (int j = 0; j < 2; ++j) b_t[j] = t_row + (((wn * 2 + j) ^ fx) << 5);

  u32x4 af[4][2], b0f[2][2], b1f[2][2];

  // fp8: per-(row, 128-k block) weight exponents (4-bit, scale 2^-e, two rows per byte) of the tile's two 128-row B groups, staged ONCE
  // per output tile behind the C staging area: LDS [2 groups][nk][64 B].  They are the first LDS-DMA copies issued for a tile, so
  // (in-order completion) they have landed whenever the first operand half-tile has.
  // The quantiser raises *sc_e_flag when any exponent is non-zero; weights whose blocks all sit within a factor 2 of their row maximum
  // (e = 0 everywhere, e.g. i.i.d. initialised weights) take the loop variant with constant block scales (wave-uniform branch).
  int esc[2][2] = {{127, 127}, {127, 127}};
  unsigned e_lane = 0, e_raw[2][2];
  const int e_shift = (frow & 1) * 4;
  bool use_exp = false;
  if constexpr (F8) {
    use_exp = g.sc_e && (!g.sc_e_flag || __builtin_amdgcn_readfirstlane(*g.sc_e_flag) != 0);
    e_lane = lds0 + LDS256 + wn * 16 + (frow >> 1);
  }
  auto load_exp = [&]() {  // exponent images of the tile `src` points at
    if constexpr (F8) {
      if (use_exp) {
        char* eimg = smem + LDS256;
        const int gp = g.sc_e_group;  // bytes per group image (multiple of 4096, >= nk * 64)
        const int grp = wave >> 2, wq = wave & 3;
        const uint8_t* es = g.sc_e + (int64_t)((grp ? s_nB1 : s_n0) >> 7) * gp;
        for (int c = 0; c < gp; c += 4096) glds16(es + c + wq * 1024 + lane * 16, eimg + grp * gp + c + wq * 1024);
      }
    }
  };

  auto k_loop = [&](auto ESC_) {
  constexpr bool ESC = decltype(ESC_)::value;
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned sb = (unsigned)(kt & 1) * STAGE256;
    const unsigned aA0 = a_row + sb + c0, aA1 = a_row + sb + c1;
    const unsigned aB0 = b_row + sb + c0, aB1 = b_row + sb + c1;
    unsigned atA[4], atB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) atA[i] = a_t[i] + sb;
#pragma unroll
    for (int j = 0; j < 2; ++j) atB[j] = b_t[j] + sb;

    // ---- phase 0: quadrant (A0, B0) ----
    if constexpr (AKS) read_frags_tr<0, 4>(af, atA); else READ_A(0);
    if constexpr (BKS) read_frags_tr<32768, 2>(b0f, atB); else READ_B(b0f, 32768);
    if constexpr (ESC) {
      const unsigned ea = e_lane + (unsigned)kt * 64;
      lds_read_u8(e_raw[0][0], ea);
      lds_read_u8(e_raw[0][1], ea + 8);
      lds_read_u8(e_raw[1][0], ea + (unsigned)g.sc_e_group);
      lds_read_u8(e_raw[1][1], ea + (unsigned)g.sc_e_group + 8);
    }
    issue(H_B1, kt + 1);
    MH_WAIT_VM(8);
    MH_BAR();
    MH_LGKM0();
    if constexpr (ESC) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) esc[b][j] = 127 - (int)((e_raw[b][j] >> e_shift) & 15u);
    }
    MFMA_QUAD(0, 0, b0f);
    MH_BAR();

    // ---- phase 1: quadrant (A0, B1) ----
    if constexpr (BKS) read_frags_tr<49152, 2>(b1f, atB); else READ_B(b1f, 49152);
    issue(H_A1, kt + 1);
    MH_WAIT_VM(8);
    MH_BAR();
    MH_LGKM0();
    MFMA_QUAD(0, 1, b1f);
    MH_BAR();

    // ---- phase 2: quadrant (A1, B1) ----
    if constexpr (AKS) read_frags_tr<16384, 4>(af, atA); else READ_A(16384);
    issue(H_A0, kt + 2);
    MH_BAR();
    MH_LGKM0();
    MFMA_QUAD(1, 1, b1f);
    MH_BAR();

    // ---- phase 3: quadrant (A1, B0): operands already in registers ----
    issue(H_B0, kt + 2);
    MH_WAIT_VM(8);
    MH_BAR();
    MFMA_QUAD(1, 0, b0f);
    MH_BAR();
  }
  };

  int v = blockIdx.x;
  setup(v);
  load_exp();
  // prologue: 6 half-tiles in consumption order; A0(0), B0(0) landed for everyone before the first read
  issue(H_A0, 0); issue(H_B0, 0); issue(H_B1, 0); issue(H_A1, 0); issue(H_A0, 1); issue(H_B0, 1);
  MH_WAIT_VM(8);
  MH_BAR();

  for (;;) {
  if (wm == 1) MH_BAR();  // the wm=1 group runs one barrier behind
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][i][b][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if constexpr (F8) {
    if (use_exp) k_loop(std::true_type{}); else k_loop(std::false_type{});
  } else {
    k_loop(std::false_type{});
  }
  if (wm == 0) MH_BAR();
  MH_WAIT_VM(0);  // the (redundant) tail loads have landed: no LDS-DMA write is in flight
  const int m0 = s_m0, n0 = s_n0, nB1 = s_nB1;  // the finished tile
  const int vn = v + gridDim.x;
  const bool more = vn < ntiles;  // (block-uniform)
  __syncthreads();  // every wave is done with the operand tiles in LDS
  if (more) {
    // the next tile's first K-tile (stage 0) is fetched under the epilogue, which stages C in [STAGE256, STAGE256 + 128 * C_ROW)
    setup(vn);
    load_exp();
    issue(H_A0, 0); issue(H_B0, 0); issue(H_B1, 0); issue(H_A1, 0);
  }
  if constexpr (F8) {  // C[m, n] = sc_m[m] * sc_n[n] * sum_k qa[m, k] qb[n, k]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float sm = g.sc_m[min(m0 + a * 128 + wm * 64 + i * 16 + (lane & 15), g.M - 1)];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int n = (b ? nB1 : n0) + wn * 32 + j * 16 + 4 * (lane >> 4);  // (fused SwiGLU: the second half-tile is the up rows)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][i][b][j][r] *= sm * g.sc_n[min(n + r, g.N - 1)];
          }
      }
  }

  // Staged epilogue (16-bit C, vectorisable layout): the accumulator layout gives a lane 4 consecutive n of one
  // row, i.e. 32-byte pieces of 16 different rows per store instruction (measured: several microseconds per
  // tile).  The block instead packs the tile, 128 rows at a time, into LDS ([128 rows][528 B]: 512 B of data + 16 B pad;
  // conflict-free for the 8-byte writes and the 16-byte reads; placed in stage 1 of the operand ring, which the prefetch
  // above does not touch) and writes it out as 16 bytes per lane = 512 contiguous bytes per row.
  if (epi_can_stage(g)) {
    char* cst = smem + STAGE256;
    const unsigned st_w = (unsigned)(wm * 64 + (lane & 15)) * C_ROW + (unsigned)(wn * 32 + 4 * (lane >> 4)) * 2;
    const int ncol = n0 + (tid & 31) * 8;
    const bool n_ok = ncol < g.N;
    auto half = [&](auto A_) {
      constexpr int a = decltype(A_)::value;
      auto fill = [&](auto EPI_) {
        constexpr int EPI = decltype(EPI_)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = min(m0 + a * 128 + wm * 64 + i * 16 + (lane & 15), g.M - 1);
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int n = min(n0 + b * 128 + wn * 32 + j * 16 + 4 * (lane >> 4), g.N - 4);
              float v4[4] = {acc[a][i][b][j][0], acc[a][i][b][j][1], acc[a][i][b][j][2], acc[a][i][b][j][3]};
              epi_xform4<DT, EPI>(g, m, n, v4);
              const uint2 pk = make_uint2(pack2<DT>(v4[0], v4[1]), pack2<DT>(v4[2], v4[3]));
              *(uint2*)(cst + st_w + (i * 16) * C_ROW + (b * 128 + j * 16) * 2) = pk;
            }
        }
      };
      switch (g.epi) {
        case 0: fill(std::integral_constant<int, 0>{}); break;
        case MH_EPI_RESIDUAL: fill(std::integral_constant<int, MH_EPI_RESIDUAL>{}); break;
        case MH_EPI_BIAS: fill(std::integral_constant<int, MH_EPI_BIAS>{}); break;
        case MH_EPI_BIAS | MH_EPI_QUICK_GELU: fill(std::integral_constant<int, MH_EPI_BIAS | MH_EPI_QUICK_GELU>{}); break;
        case MH_EPI_BIAS | MH_EPI_RESIDUAL: fill(std::integral_constant<int, MH_EPI_BIAS | MH_EPI_RESIDUAL>{}); break;
        default: break;  // excluded by epi_can_stage
      }
      __syncthreads();
      const int mh = m0 + a * 128;  // first global row of this half
      if (g.sw_mode == 1) {
        // fused SwiGLU forward: LDS columns 0-127 = gate [n0, n0+128), 128-255 = up ff + [n0, n0+128) (16-bit, rounded)
        const int c = tid & 31;
        const int col = n0 + (c & 15) * 8;  // gate / act column of this chunk
        if (col < g.sw_ff) {
#pragma unroll 2
          for (int pass = 0; pass < 8; ++pass) {
            const int row = pass * 16 + (tid >> 5);
            if (mh + row >= g.M) continue;
            const uint4 mine = *(const uint4*)(cst + row * C_ROW + c * 16);
            uint16_t* gu = (uint16_t*)g.C + (int64_t)(mh + row) * g.ldc;
            if (c < 16) {
              *(uint4*)(gu + col) = mine;
              float ga[8], ub[8];
              unpack8<DT>(mine, ga);
              unpack8<DT>(*(const uint4*)(cst + row * C_ROW + (c + 16) * 16), ub);
#pragma unroll
              for (int k = 0; k < 8; ++k) ga[k] = swiglu_fwd1(ga[k], ub[k]);
              *(uint4*)((uint16_t*)g.sw_out + (int64_t)(mh + row) * g.sw_ldo + col) = pack8<DT>(ga);
            } else {
              *(uint4*)(gu + g.sw_ff + col) = mine;
            }
          }
        }
      } else if (g.sw_mode == 2) {
        // fused SwiGLU backward: the staged tile is dact (rounded to 16 bits exactly as the unfused path stores it)
        if (n_ok) {
#pragma unroll 2
          for (int pass = 0; pass < 8; ++pass) {
            const int row = pass * 16 + (tid >> 5);
            if (mh + row >= g.M) continue;
            float d_[8], ga[8], ub[8], dg[8], du[8];
            unpack8<DT>(*(const uint4*)(cst + row * C_ROW + (tid & 31) * 16), d_);
            const uint16_t* gu = (const uint16_t*)g.sw_in + (int64_t)(mh + row) * g.sw_ldi;
            unpack8<DT>(*(const uint4*)(gu + ncol), ga);
            unpack8<DT>(*(const uint4*)(gu + g.sw_ff + ncol), ub);
#pragma unroll
            for (int k = 0; k < 8; ++k) swiglu_bwd1(ga[k], ub[k], d_[k], dg[k], du[k]);
            uint16_t* dgu = (uint16_t*)g.sw_out + (int64_t)(mh + row) * g.sw_ldo;
            *(uint4*)(dgu + ncol) = pack8<DT>(dg);
            *(uint4*)(dgu + g.sw_ff + ncol) = pack8<DT>(du);
          }
        }
      } else if (g.sw_mode == 3) {
        // fused quick-GELU forward (CLIP MLP, fc1): the staged tile is f1 = x W1^T + b1 rounded to 16 bits; C = f1 (the backward needs it) and
        // sw_out = a = f1 * sigmoid(1.702 f1) from the rounded values, exactly as mh_quick_gelu_fwd computes it from the stored tensor
        if (n_ok) {
#pragma unroll 2
          for (int pass = 0; pass < 8; ++pass) {
            const int row = pass * 16 + (tid >> 5);
            if (mh + row >= g.M) continue;
            const uint4 val = *(const uint4*)(cst + row * C_ROW + (tid & 31) * 16);
            *(uint4*)((uint16_t*)g.C + (int64_t)(mh + row) * g.ldc + ncol) = val;
            float x[8];
            unpack8<DT>(val, x);
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = x[k] * sigmoidf_(1.702f * x[k]);
            *(uint4*)((uint16_t*)g.sw_out + (int64_t)(mh + row) * g.sw_ldo + ncol) = pack8<DT>(x);
          }
        }
      } else if (g.sw_mode == 4) {
        // fused quick-GELU backward (fc2 dgrad): the staged tile is da = dY W2 rounded to 16 bits and never stored; sw_out = df1 =
        // da * s * (1 + 1.702 f1 (1 - s)), s = sigmoid(1.702 f1), with f1 = sw_in - mh_quick_gelu_bwd's arithmetic on the same values
        if (n_ok) {
#pragma unroll 2
          for (int pass = 0; pass < 8; ++pass) {
            const int row = pass * 16 + (tid >> 5);
            if (mh + row >= g.M) continue;
            float z[8], x[8];
            unpack8<DT>(*(const uint4*)(cst + row * C_ROW + (tid & 31) * 16), z);
            unpack8<DT>(*(const uint4*)((const uint16_t*)g.sw_in + (int64_t)(mh + row) * g.sw_ldi + ncol), x);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const float sg = sigmoidf_(1.702f * x[k]);
              x[k] = z[k] * sg * (1.0f + 1.702f * x[k] * (1.0f - sg));
            }
            *(uint4*)((uint16_t*)g.sw_out + (int64_t)(mh + row) * g.sw_ldo + ncol) = pack8<DT>(x);
          }
        }
      } else if (g.rope_tab && n0 < g.rope_cols) {
        // fused RoPE (llama_flash_attn_monkey_patch.py:56-59): the tile holds whole heads (256 % D == 0, rope_cols % D == 0), so the
        // thread that owns a low-half 8-channel chunk also reads its partner chunk D/2 channels later from the same LDS row and
        // rotates the two ROUNDED 16-bit values exactly as the stand-alone mh_rope_qk does on the stored tensor; chunks at or
        // beyond rope_cols (the v heads of a tile that straddles the boundary) are copied through.
        const int hc = g.rope_D >> 4;            // chunks per half head
        const int c = tid & 31;
        const bool rot = ncol < g.rope_cols;
        if (!rot || (c % (2 * hc)) < hc) {
          const int j0 = (c % hc) * 8;           // first rotary pair index of this chunk
#pragma unroll 2
          for (int pass = 0; pass < 8; ++pass) {
            const int row = pass * 16 + (tid >> 5);
            if (mh + row >= g.M) continue;
            uint16_t* dst = (uint16_t*)g.C + (int64_t)(mh + row) * g.ldc + ncol;
            if (!rot) {
              if (n_ok) *(uint4*)dst = *(const uint4*)(cst + row * C_ROW + c * 16);
              continue;
            }
            float lo[8], hi[8];
            unpack8<DT>(*(const uint4*)(cst + row * C_ROW + c * 16), lo);
            unpack8<DT>(*(const uint4*)(cst + row * C_ROW + (c + hc) * 16), hi);
            const float2* tb = (const float2*)g.rope_tab + (int64_t)((mh + row) % g.rope_S) * (g.rope_D >> 1) + j0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              rope_rot(lo[k], hi[k], tb[k].x, tb[k].y, lo[k], hi[k]);
            }
            *(uint4*)dst = pack8<DT>(lo);
            *(uint4*)(dst + (g.rope_D >> 1)) = pack8<DT>(hi);
          }
        }
      } else {
#pragma unroll 4
        for (int pass = 0; pass < 8; ++pass) {
          const int row = pass * 16 + (tid >> 5);
          const uint4 val = *(const uint4*)(cst + row * C_ROW + (tid & 31) * 16);
          if (n_ok && mh + row < g.M) *(uint4*)((uint16_t*)g.C + (int64_t)(mh + row) * g.ldc + ncol) = val;
        }
      }
    };
    half(std::integral_constant<int, 0>{});
    __syncthreads();  // the first half has been read out of the staging rows
    half(std::integral_constant<int, 1>{});
  } else {
    epi_dispatch<DT>(g, [&](auto store) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = m0 + a * 128 + wm * 64 + i * 16 + (lane & 15);
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int n = n0 + b * 128 + wn * 32 + j * 16 + 4 * (lane >> 4);
              const f32x4_t val = acc[a][i][b][j];
              store(m, n, val[0], val[1], val[2], val[3]);
            }
        }
    });
  }
  if (!more) break;
  v = vn;
  MH_WAIT_VM(0);    // stage 0 of the next tile has landed (it had the whole epilogue); this tile's stores are acknowledged
  __syncthreads();  // ... for every thread, and the staging rows (stage 1) are free again
  issue(H_A0, 1); issue(H_B0, 1);
  }
}

}  // namespace

// Grid of a launch: one block per CU looping over the output tiles (persistent, default) or one block per tile.
int g_persistent = 1;
static int grid_x(const GemmArgs& g) {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    ncu &= ~7;  // a multiple of the 8 XCDs: a block's tiles all map to its own XCD's run (tile_of)
    if (ncu <= 0) ncu = 8;
  }
  const int ntiles = g.tiles_m * g.tiles_n;
  return g_persistent ? min(ntiles, ncu) : ntiles;
}

template <int DT, bool AKS, bool BKS>
int launch_one(const GemmArgs& g, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_nt_256<DT, AKS, BKS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS256);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_nt_256<DT, AKS, BKS>), dim3(grid_x(g), g.splits), dim3(512), LDS256, stream, g);
  MH_LAUNCH_CHECK();
}

int launch_gemm_256(const GemmArgs& g, int dt, int a_kstrided, int b_kstrided, hipStream_t stream) {
  const int key = (dt == MH_BF16 ? 0 : 4) | (a_kstrided ? 2 : 0) | (b_kstrided ? 1 : 0);
  switch (key) {
    case 0: return launch_one<MH_BF16, false, false>(g, stream);
    case 1: return launch_one<MH_BF16, false, true>(g, stream);
    case 2: return launch_one<MH_BF16, true, false>(g, stream);
    case 3: return launch_one<MH_BF16, true, true>(g, stream);
    case 4: return launch_one<MH_F16, false, false>(g, stream);
    case 5: return launch_one<MH_F16, false, true>(g, stream);
    case 6: return launch_one<MH_F16, true, false>(g, stream);
    default: return launch_one<MH_F16, true, true>(g, stream);
  }
}

int launch_gemm_nt_256(const GemmArgs& g, int dt, hipStream_t stream) { return launch_gemm_256(g, dt, 0, 0, stream); }

template <int DT>
int launch_f8(const GemmArgs& g, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_nt_256<DT, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS256 + F8_EXP_BYTES);
    attr_set = true;
  }
  if (g.sc_e && (2 * g.sc_e_group > F8_EXP_BYTES || (g.sc_e_group & 4095) || g.sc_e_group < ((g.K * 2 + 127) / 128) * 64)) return MH_ERR_SHAPE;
  hipLaunchKernelGGL((gemm_nt_256<DT, false, false, true>), dim3(grid_x(g), 1), dim3(512), LDS256 + (g.sc_e ? 2 * g.sc_e_group : 0), stream, g);
  MH_LAUNCH_CHECK();
}
// fp8 operands: g.A / g.B point at bytes, g.K, g.lda, g.ldb are in 2-BYTE units (K/2 etc.), dt = output type
int launch_gemm_nt_256_f8(const GemmArgs& g, int dt, hipStream_t stream) {
  return dt == MH_BF16 ? launch_f8<MH_BF16>(g, stream) : launch_f8<MH_F16>(g, stream);
}

}  // namespace mhgemm

extern "C" void mh_gemm_persistent(int on) { mhgemm::g_persistent = on ? 1 : 0; }
