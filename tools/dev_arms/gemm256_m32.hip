// gemm_nt_256 with MFMA 32x32x16 fragments (experiment / A-B arm of gemm256.hip; same pipeline, same tile).
// A wave's 64x32 quadrant is 2 x 1 blocks of 32x32, 4 k-steps of 16 per K-tile = 8 MFMA 32x32x16 per phase
// (half the MFMA instruction count of the 16x16x32 arm at identical FLOPs, operands and LDS traffic).
// Fragment of a K-contiguous half-tile: lane l holds row (l&31), 16-byte chunk 2*ks + (l>>5) of that row.
#include "gemm_common.h"

namespace mhgemm {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int DT>
__device__ __forceinline__ f32x16_t mfma32x(u32x4 a, u32x4 b, f32x16_t c) {
  if constexpr (DT == MH_BF16)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

constexpr int HALF_BYTES = 128 * BK * 2;
constexpr int STAGE256 = 4 * HALF_BYTES;
constexpr int H_A0 = 0, H_A1 = 1, H_B0 = 2, H_B1 = 3;

#define M32_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define M32_BAR()                        \
  do {                                   \
    __builtin_amdgcn_sched_barrier(0);   \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)
#define M32_LGKM0()                                       \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);                    \
  } while (0)
#define DSR32(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
// A half-tile at BASE: 2 row blocks (4096 B apart) x 4 k-steps (address regs aA[ks])
#define M32_READ_A(BASE)                                                                          \
  do {                                                                                            \
    DSR32(af[0][0], aA[0], BASE + 0);    DSR32(af[0][1], aA[1], BASE + 0);                        \
    DSR32(af[0][2], aA[2], BASE + 0);    DSR32(af[0][3], aA[3], BASE + 0);                        \
    DSR32(af[1][0], aA[0], BASE + 4096); DSR32(af[1][1], aA[1], BASE + 4096);                     \
    DSR32(af[1][2], aA[2], BASE + 4096); DSR32(af[1][3], aA[3], BASE + 4096);                     \
  } while (0)
#define M32_READ_B(bf, BASE)                                                                      \
  do {                                                                                            \
    DSR32(bf[0], aB[0], BASE); DSR32(bf[1], aB[1], BASE); DSR32(bf[2], aB[2], BASE); DSR32(bf[3], aB[3], BASE); \
  } while (0)
#define M32_QUAD(MH, NH, bf)                                                                      \
  do {                                                                                            \
    __builtin_amdgcn_s_setprio(1);                                                                \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                              \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                               \
        acc[MH][i][NH] = mfma32x<DT>(bf[ks], af[i][ks], acc[MH][i][NH]);                          \
    __builtin_amdgcn_s_setprio(0);                                                                \
  } while (0)

template <int DT>
__global__ __launch_bounds__(512, 2) void gemm_nt_256_m32(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  int tm, tn;
  tile_of_block(g, tm, tn);
  const int m0 = tm * 256, n0 = tn * 256;
  const int nk = g.K / BK;

  const uint16_t* src[4][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int qd = i * 512 + tid;
    const int row = qd >> 3, cc = qd & 7;
    const int c = (cc ^ ((row >> 1) & 7)) * 8;
    src[H_A0][i] = g.A + (int64_t)min(m0 + row, g.M - 1) * g.lda + c;
    src[H_A1][i] = g.A + (int64_t)min(m0 + 128 + row, g.M - 1) * g.lda + c;
    src[H_B0][i] = g.B + (int64_t)min(n0 + row, g.N - 1) * g.ldb + c;
    src[H_B1][i] = g.B + (int64_t)min(n0 + 128 + row, g.N - 1) * g.ldb + c;
  }
  auto issue = [&](int h, int kt) {
    const int koff = min(kt, nk - 1) * BK;
    char* dst = smem + (kt & 1) * STAGE256 + h * HALF_BYTES + wave * 1024;
    glds16(src[h][0] + koff, dst);
    glds16(src[h][1] + koff, dst + 8192);
  };

  f32x16_t acc[2][2][2];  // [mh][i][nh]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][i][b][r] = 0.f;

  const int l31 = lane & 31, hi = lane >> 5;
  const int swz = (l31 >> 1) & 7;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned a_row = lds0 + (wm * 64 + l31) * 128;
  const unsigned b_row = lds0 + (wn * 32 + l31) * 128;
  unsigned coff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) coff[ks] = ((2 * ks + hi) ^ swz) << 4;

  u32x4 af[2][4], b0f[4], b1f[4];

  issue(H_A0, 0); issue(H_B0, 0); issue(H_B1, 0); issue(H_A1, 0); issue(H_A0, 1); issue(H_B0, 1);
  M32_WAIT_VM(8);
  M32_BAR();
  if (wm == 1) M32_BAR();

  for (int kt = 0; kt < nk; ++kt) {
    const unsigned sb = (unsigned)(kt & 1) * STAGE256;
    unsigned aA[4], aB[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      aA[ks] = a_row + sb + coff[ks];
      aB[ks] = b_row + sb + coff[ks];
    }
    // phase 0: (A0, B0)
    M32_READ_A(0);
    M32_READ_B(b0f, 32768);
    issue(H_B1, kt + 1);
    M32_WAIT_VM(8);
    M32_BAR();
    M32_LGKM0();
    M32_QUAD(0, 0, b0f);
    M32_BAR();
    // phase 1: (A0, B1)
    M32_READ_B(b1f, 49152);
    issue(H_A1, kt + 1);
    M32_WAIT_VM(8);
    M32_BAR();
    M32_LGKM0();
    M32_QUAD(0, 1, b1f);
    M32_BAR();
    // phase 2: (A1, B1)
    M32_READ_A(16384);
    issue(H_A0, kt + 2);
    M32_BAR();
    M32_LGKM0();
    M32_QUAD(1, 1, b1f);
    M32_BAR();
    // phase 3: (A1, B0)
    issue(H_B0, kt + 2);
    M32_WAIT_VM(8);
    M32_BAR();
    M32_QUAD(1, 0, b0f);
    M32_BAR();
  }
  if (wm == 0) M32_BAR();
  M32_WAIT_VM(0);

  // D = Bfrag x Afrag: lane holds m = l31, n = (r&3) + 8*(r>>2) + 4*hi  -> 4 consecutive n per register quad
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + a * 128 + wm * 64 + i * 32 + l31;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + b * 128 + wn * 32 + 8 * q + 4 * hi;
          const f32x16_t v = acc[a][i][b];
          epi_store4<DT>(g, m, n, v[4 * q + 0], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        }
    }
}


// ---- second variant: TWO phases per K-tile, cut along the A halves instead of four quadrants ---------------------------------
// The quadrant phases above give a phase two 32x32 accumulators with a 4-deep dependent chain each (acc, k-step) - the form that
// measured 10 % slower than 16x16x32.  Here phase 0 is (A0 x [B0 | B1]) and phase 1 is (A1 x [B0 | B1]): four accumulators per phase,
// visited round-robin, so two MFMAs on one accumulator are four instructions apart; B fragments (both halves) live in registers
// for the whole K-tile.  Staging: the same half-tile slots and the same one-barrier stagger of the wm groups; a half-tile is
// requested in the MFMA segment two phases before the segment that waits for it (see the schedule at the loop).
#define M32K_PHASE(MH)                                                                            \
  do {                                                                                            \
    __builtin_amdgcn_s_setprio(1);                                                                \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                              \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                             \
        acc[MH][i][0] = mfma32x<DT>(b0f[ks], af[i][ks], acc[MH][i][0]);                           \
        acc[MH][i][1] = mfma32x<DT>(b1f[ks], af[i][ks], acc[MH][i][1]);                           \
      }                                                                                           \
    __builtin_amdgcn_s_setprio(0);                                                                \
  } while (0)

template <int DT>
__global__ __launch_bounds__(512, 2) void gemm_nt_256_m32k(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  int tm, tn;
  tile_of_block(g, tm, tn);
  const int m0 = tm * 256, n0 = tn * 256;
  const int nk = g.K / BK;

  const uint16_t* src[4][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int qd = i * 512 + tid;
    const int row = qd >> 3, cc = qd & 7;
    const int c = (cc ^ ((row >> 1) & 7)) * 8;
    src[H_A0][i] = g.A + (int64_t)min(m0 + row, g.M - 1) * g.lda + c;
    src[H_A1][i] = g.A + (int64_t)min(m0 + 128 + row, g.M - 1) * g.lda + c;
    src[H_B0][i] = g.B + (int64_t)min(n0 + row, g.N - 1) * g.ldb + c;
    src[H_B1][i] = g.B + (int64_t)min(n0 + 128 + row, g.N - 1) * g.ldb + c;
  }
  auto issue = [&](int h, int kt) {
    const int koff = min(kt, nk - 1) * BK;
    char* dst = smem + (kt & 1) * STAGE256 + h * HALF_BYTES + wave * 1024;
    glds16(src[h][0] + koff, dst);
    glds16(src[h][1] + koff, dst + 8192);
  };

  f32x16_t acc[2][2][2];  // [mh][i][nh]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][i][b][r] = 0.f;

  const int l31 = lane & 31, hi = lane >> 5;
  const int swz = (l31 >> 1) & 7;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned a_row = lds0 + (wm * 64 + l31) * 128;
  const unsigned b_row = lds0 + (wn * 32 + l31) * 128;
  unsigned coff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) coff[ks] = ((2 * ks + hi) ^ swz) << 4;

  u32x4 af[2][4], b0f[4], b1f[4];

  // requests, in the order they are waited for: [A0 B0 B1](0), A1(0), [A0 B0 B1](1); then per K-tile t: A1(t+1) in the MFMA segment of
  // phase 0, [A0 B0 B1](t+2) in the MFMA segment of phase 1.  Waits (own loads; the barrier after them publishes the half-tiles):
  // read segment of phase 0 waits for A1(t): 6 newer loads may stay in flight; of phase 1 for [A0 B0 B1](t+1): 2 newer.
  issue(H_A0, 0); issue(H_B0, 0); issue(H_B1, 0); issue(H_A1, 0); issue(H_A0, 1); issue(H_B0, 1); issue(H_B1, 1);
  M32_WAIT_VM(8);
  M32_BAR();
  if (wm == 1) M32_BAR();

  for (int kt = 0; kt < nk; ++kt) {
    const unsigned sb = (unsigned)(kt & 1) * STAGE256;
    unsigned aA[4], aB[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      aA[ks] = a_row + sb + coff[ks];
      aB[ks] = b_row + sb + coff[ks];
    }
    // phase 0: A0 x (B0 | B1)
    M32_READ_A(0);
    M32_READ_B(b0f, 32768);
    M32_READ_B(b1f, 49152);
    M32_WAIT_VM(6);
    M32_BAR();
    issue(H_A1, kt + 1);
    M32_LGKM0();
    M32K_PHASE(0);
    M32_BAR();
    // phase 1: A1 x (B0 | B1)
    M32_READ_A(16384);
    M32_WAIT_VM(2);
    M32_BAR();
    issue(H_A0, kt + 2); issue(H_B0, kt + 2); issue(H_B1, kt + 2);
    M32_LGKM0();
    M32K_PHASE(1);
    M32_BAR();
  }
  if (wm == 0) M32_BAR();
  M32_WAIT_VM(0);

#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + a * 128 + wm * 64 + i * 32 + l31;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + b * 128 + wn * 32 + 8 * q + 4 * hi;
          const f32x16_t v = acc[a][i][b];
          epi_store4<DT>(g, m, n, v[4 * q + 0], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        }
    }
}

}  // namespace

int g_m32_kcut = 0;  // 1: the two-phase (K-cut) variant
int launch_gemm_nt_256_m32(const GemmArgs& g, int dt, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_nt_256_m32<MH_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE256);
    hipFuncSetAttribute((const void*)gemm_nt_256_m32<MH_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE256);
    attr_set = true;
  }
  const int grid = g.tiles_m * g.tiles_n;
  if (g_m32_kcut) {
    static bool attr2 = false;
    if (!attr2) {
      hipFuncSetAttribute((const void*)gemm_nt_256_m32k<MH_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE256);
      hipFuncSetAttribute((const void*)gemm_nt_256_m32k<MH_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE256);
      attr2 = true;
    }
    if (dt == MH_BF16) hipLaunchKernelGGL(gemm_nt_256_m32k<MH_BF16>, dim3(grid), dim3(512), 2 * STAGE256, stream, g);
    else hipLaunchKernelGGL(gemm_nt_256_m32k<MH_F16>, dim3(grid), dim3(512), 2 * STAGE256, stream, g);
    MH_LAUNCH_CHECK();
  }
  if (dt == MH_BF16)
    hipLaunchKernelGGL(gemm_nt_256_m32<MH_BF16>, dim3(grid), dim3(512), 2 * STAGE256, stream, g);
  else
    hipLaunchKernelGGL(gemm_nt_256_m32<MH_F16>, dim3(grid), dim3(512), 2 * STAGE256, stream, g);
  MH_LAUNCH_CHECK();
}

}  // namespace mhgemm
