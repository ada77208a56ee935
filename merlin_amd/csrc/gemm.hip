// bf16/f16 MFMA GEMM, C[M,N] = A[M,K] * B[N,K]^T with fused epilogues (gfx950).
//
// Common to both kernels
//   * both operands are K-contiguous ("NT"): a tile row is 64 k = 128 B = 8 x 16-B chunks.
//   * global -> LDS by `global_load_lds_dwordx4` (no VGPR round trip).  The LDS image must be lane-linear
//     (hardware requirement), so the bank swizzle is applied on the per-lane GLOBAL source address and
//     again on the ds_read address (cdna_hip_programming.md rule 21): chunk' = chunk ^ ((row >> 1) & 7).
//     With 128-B rows two rows share one 256-B bank row, so a ds_read_b128 lane group (16 lanes, 16
//     consecutive rows at one logical chunk) lands on 16 distinct 16-B slots: conflict-free.
//   * MFMA 16x16x32, operands passed swapped (B-fragment as the A operand) so each lane ends up with 4
//     CONSECUTIVE n for one m: the epilogue reads bias/residual and writes C as 8-byte (16-bit C) or
//     16-byte (fp32 C) vectors.
//   * block id -> tile: XCD-aware remap (block b runs on XCD b%8; each XCD gets a contiguous tile range,
//     bijective for any grid) + grouped ordering (8 m-tiles per group) so co-resident blocks of one XCD
//     share A/B panels in that XCD's private L2.
//   * M and N edges: loads clamp the row index, stores are predicated.  K % 64 == 0 is required.
//
// gemm_nt_256 (large problems): 256x256x64 block tile, 512 threads = 8 waves (2 x 4), each wave 128x64
//   of the output held as 32 accumulator fragments (128 VGPRs).  A K-tile is four 16-KiB half-tiles
//   (A rows 0-127 / 128-255, B rows 0-127 / 128-255) in one of two LDS stages (128 KiB total).  A wave
//   owns rows {wm*64..+64} of EACH A half and columns {wn*32..+32} of EACH B half, so the four output
//   quadrants of a K-tile are visited as (A0,B0) (A0,B1) (A1,B1) (A1,B0): four phases of 16 MFMAs, with
//   fragment reads A0+B0 | B1 | A1 | none.  Each LDS half-tile is therefore dead after phase 0 / 0 / 1 /
//   2 and is refilled RIGHT THEN with the half-tile of K-tile t+2 (t+1 for A1): one half-tile (two
//   global_load_lds per thread) is issued per phase, every load has >= 6 phases (~1.5 K-tiles) to land,
//   and the only wait is a COUNTED `s_waitcnt vmcnt(10)` (five half-tiles stay in flight) in front of a
//   raw s_barrier at the start of phases 0-2; phase 3 needs neither.  Never vmcnt(0) in the loop.
// gemm_nt_128 (small problems / few tiles): 128x128x64 tile, 4 waves, double buffer, one barrier per
//   K-tile with a full drain (the "2-phase" structure, ~800-900 TF).
#include "mh_common.h"
#include "gemm_common.h"

using namespace mhgemm;

namespace {

// ---------------------------------------------------------------------------------------------------
// 128 x 128 x 64, 4 waves
// ---------------------------------------------------------------------------------------------------
constexpr int BM128 = 128, BN128 = 128;
constexpr int STAGE128 = (BM128 + BN128) * BK * 2;  // 32 KiB
constexpr int A128_BYTES = BM128 * BK * 2;          // 16 KiB

template <int DT>
__global__ __launch_bounds__(256, 2) void gemm_nt_128(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int tm, tn;
  tile_of_block(g, tm, tn);
  const int m0 = tm * BM128, n0 = tn * BN128;

  const uint16_t* asrc[4];
  const uint16_t* bsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int qd = i * 256 + tid;
    const int row = qd >> 3, cc = qd & 7;
    const int c = cc ^ ((row >> 1) & 7);
    const int gm = min(m0 + row, g.M - 1);
    const int gn = min(n0 + row, g.N - 1);
    asrc[i] = g.A + (int64_t)gm * g.lda + c * 8;
    bsrc[i] = g.B + (int64_t)gn * g.ldb + c * 8;
  }
  const int nk = g.K / BK;

  auto stage = [&](int s, int kt) {
    char* sA = smem + s * STAGE128;
    char* sB = sA + A128_BYTES;
    const int koff = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(asrc[i] + koff, sA + (i * 256 + wave * 64) * 16);
      glds16(bsrc[i] + koff, sB + (i * 256 + wave * 64) * 16);
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int swz = (frow >> 1) & 7;
  const int kq = lane >> 4;
  const int a_off = (wm * 64 + frow) * 128;
  const int b_off = (wn * 64 + frow) * 128;

  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile kt landed for everyone; everyone is done reading the other stage
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* sA = smem + (kt & 1) * STAGE128;
    const char* sB = sA + A128_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + kq) ^ swz) << 4;
      uint4 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *(const uint4*)(sA + a_off + i * 16 * 128 + coff);
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = *(const uint4*)(sB + b_off + j * 16 * 128 + coff);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<DT>(bf[j], af[i], acc[i][j]);
    }
  }

  // lane holds C[m = .. + (lane&15)][n = .. + 4*(lane>>4) + r], r = 0..3
  epi_dispatch<DT>(g, [&](auto store) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
        store(m, n, acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
    }
  });
}


// out[C, R_pad] = in[R, C]^T, 16-bit elements, 64x64 tiles through LDS; columns [R, R_pad) are zero filled
// (the transposed operand's K dimension must be a multiple of 64 for mh_gemm_nt).
__global__ __launch_bounds__(256) void transpose16_kernel(const uint16_t* __restrict__ in, int64_t ldi,
                                                          uint16_t* __restrict__ out, int64_t ldo, int R, int C, int R_pad) {
  __shared__ uint16_t t[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    t[i][tx] = (r < R && c < C) ? in[(int64_t)r * ldi + c] : (uint16_t)0;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < R_pad) out[(int64_t)c * ldo + r] = t[tx][i];
  }
}

int g_gemm_gm = 0;     // raster group height (tile rows); 0 = automatic: 4 (measured 4 < 8 < 16 on the decoder shapes, profiles/r02_gemm_raster_ab.txt), or ALL tile rows when
                       // there are at most 8 of them (a short prefill: a group of 4 leaves the 5th 128-row tile row of a 613-token sequence in a group of its own, whose
                       // tiles re-read every weight panel on other XCDs; cfg-2 forward 14.59 -> 14.38 ms same box, profiles/r06_cfg2_raster_ab.txt)
int g_force_kernel = 0;  // 0 auto, 128, 256 (8 waves), 4 (4 waves), 5.. (timing probes) - tests / A-B benchmarking

}  // namespace

extern "C" void mh_gemm_force_kernel(int which) { g_force_kernel = which; }
extern "C" void mh_gemm_raster_group(int gm) { g_gemm_gm = gm >= 1 && gm <= 64 ? gm : 0; }

extern "C" int mh_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                          const void* bias, const void* resid, int64_t ldr, int M, int N, int K, int dt,
                          int epilogue, void* stream) {
  return mh_gemm(A, lda, 0, B, ldb, 0, C, ldc, bias, resid, ldr, M, N, K, dt, epilogue, stream);
}

__device__ uint16_t g_zero_row[512];  // zero-initialised: stands in for K-strided rows k >= K (see gemm256.hip)

// sum of `splits` fp32 partial tiles -> 16-bit or fp32 C (+ old C): 4 consecutive elements per thread
template <int DT>
__global__ __launch_bounds__(256) void splitk_reduce_k(const float* __restrict__ ws, int splits, int64_t split_stride, void* C,
                                                       int64_t ldc, int M, int N, int out_f32, int accumulate) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;  // quad index over M * N/4
  const int nq = N >> 2;
  if (q >= (int64_t)M * nq) return;
  const int m = (int)(q / nq), n = (int)(q % nq) * 4;
  float4 s = *(const float4*)(ws + (int64_t)m * N + n);
  for (int i = 1; i < splits; ++i) {
    const float4 t = *(const float4*)(ws + i * split_stride + (int64_t)m * N + n);
    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
  }
  if (out_f32) {
    float4* dst = (float4*)((float*)C + (int64_t)m * ldc + n);
    if (accumulate) { const float4 o = *dst; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    *dst = s;
  } else {
    uint2* dst = (uint2*)((uint16_t*)C + (int64_t)m * ldc + n);
    if (accumulate) {
      const uint2 o = *dst;
      float o0, o1, o2, o3;
      unpack2<DT>(o.x, o0, o1);
      unpack2<DT>(o.y, o2, o3);
      s.x += o0; s.y += o1; s.z += o2; s.w += o3;
    }
    *dst = make_uint2(pack2<DT>(s.x, s.y), pack2<DT>(s.z, s.w));
  }
}

// the same with the GEMM's own epilogue (bias, quick-GELU, residual, accumulate, fp32 store: epi_store4) behind the sum: split-K for
// the skinny forward / dgrad products of short sequences (mh_gemm_splitk_epi)
template <int DT>
__global__ __launch_bounds__(256) void splitk_reduce_epi_k(const float* __restrict__ ws, int splits, int64_t split_stride, GemmArgs g) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;  // quad index over M * N/4
  const int nq = g.N >> 2;
  if (q >= (int64_t)g.M * nq) return;
  const int m = (int)(q / nq), n = (int)(q % nq) * 4;
  float4 s = *(const float4*)(ws + (int64_t)m * g.N + n);
  for (int i = 1; i < splits; ++i) {
    const float4 t = *(const float4*)(ws + i * split_stride + (int64_t)m * g.N + n);
    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
  }
  epi_store4<DT>(g, m, n, s.x, s.y, s.z, s.w);
}

struct RopeSpec {
  const float* tab = nullptr; int S = 0, D = 0, cols = 0;
  int sw_mode = 0, sw_ff = 0; void* sw_out = nullptr; const void* sw_in = nullptr; int64_t sw_ldo = 0, sw_ldi = 0;  // fused SwiGLU
};
// Shapes that go to gemm_w4 by default (see the call site).  g_w4_mask: bit 0 = TN (wgrad, incl. its split-K form), bit 1 = NN (dgrad),
// bit 2 = plain / residual NT, bit 3 = the fp8 NT products of the fp8 training step, bit 4 = the fused forward forms (q|k|v + RoPE,
// gate|up + SwiGLU: rotated / gated on the fp32 accumulators), bit 5 = the SwiGLU-backward dgrad, bit 6 = fp32 outputs of NT products
// (lm_head logits, the fp32 residual streams' accumulating projections), bit 7 = the fused forward forms at any size (see the call site),
// bit 8 = the fused forms of the fp8 training step (RoPE, SwiGLU forward / backward, fp32 logits) on the 4-wave fp8 kernel.
// Measured per bit in the step: profiles/r04_gemm_w4_policy.txt.
int g_w4_mask = 3 | 8 | 16 | 32 | 64 | 128 | 256;  // (everything: with the prefetching epilogues every form measures at or above the 8-wave kernel)
extern "C" void mh_gemm_w4_policy(int mask) { g_w4_mask = mask; }
// 128-row block tiles of the 4-wave kernel (NT products): 0 = never, 1 = where the launch plan below says they take fewer rounds (default),
// 2 = wherever the kernel exists (tests / A-B)
int g_w4_half = 1;
extern "C" void mh_gemm_w4_half(int mode) { g_w4_half = mode; }
// Rounds of the 256 CUs a product takes in 256-row tiles against 128-row tiles (one block per CU).  A round of half tiles is priced at 0.75 of
// a round of full ones - measured (profiles/r05_w4_half_ab.txt: 60-67 us against 81-87 for 64 K-tiles): a K-tile of a half tile is half the
// MFMAs but 48 KiB of operand copies instead of 64, and it is the copies that set the pace of either (~48 GB/s into one CU's LDS).  A 613-token
// prefill: q|k|v 144 tiles = 1 round vs 240 half tiles = 0.75; gate|up 258 = 2 rounds vs 430 = 1.5; a training batch of 32 768 tokens never
// gets here (24 vs 36).
static bool w4_half_pays(int M, int tiles_n, int splits) {
  const int64_t t256 = (int64_t)((M + 255) / 256) * tiles_n * splits, t128 = (int64_t)((M + 127) / 128) * tiles_n * splits;
  const double r256 = (double)((t256 + 255) / 256), r128 = 0.75 * (double)((t128 + 255) / 256);
  return r128 < r256 - 1e-9;
}
static bool w4_policy(int a_ks, int b_ks, int M, int N, int K, int epi, const RopeSpec& fx) {
  (void)M; (void)N;
  if (K < 4096) return false;
  if (fx.sw_mode == 1 || fx.tab) return (g_w4_mask & 16) != 0;
  if (fx.sw_mode == 2) return (g_w4_mask & 32) != 0;
  const int form = (a_ks && b_ks) ? 1 : (b_ks ? 2 : (a_ks ? 0 : 4));
  if (epi & MH_EPI_OUT_F32) {
    if (form == 1) return (g_w4_mask & 1) != 0;   // split-K partials of a weight gradient
    // plain fp32 store (lm_head: +4.3 %) and the accumulating store into an fp32 residual stream (old values requested 16 tiles ahead:
    // o projection +3.4 %, down projection +4.4 %) - profiles/r04_w4_forms_ab.txt
    return form == 4 && (g_w4_mask & 64) != 0;
  }
  if (!(g_w4_mask & form)) return false;
  if (form == 4) return (epi & ~MH_EPI_ACCUM) == 0 || (epi & ~MH_EPI_ACCUM) == MH_EPI_RESIDUAL;
  return (epi & ~MH_EPI_ACCUM) == 0;
}

static int gemm_impl(const void* A, int64_t lda, int a_kstrided, const void* B, int64_t ldb, int b_kstrided, void* C,
                     int64_t ldc, const void* bias, const void* resid, int64_t ldr, int M, int N, int K, int dt,
                     int epilogue, int splits, int64_t c_split, void* stream, RopeSpec rope = RopeSpec());

extern "C" int mh_gemm_nt_rope(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                               int dt, const float* cos_sin, int S, int D, int rope_cols, void* stream) {
  // staged-epilogue conditions + whole heads per 256-column tile
  if (!cos_sin || S <= 0 || (D != 128 && D != 64) || rope_cols <= 0 || rope_cols > N || (rope_cols % D) || (N & 7) || (ldc & 7) ||
      ((((uintptr_t)C) & 15u) != 0))
    return MH_ERR_ARG;
  RopeSpec r;
  r.tab = cos_sin; r.S = S; r.D = D; r.cols = rope_cols;
  return gemm_impl(A, lda, 0, B, ldb, 0, C, ldc, nullptr, nullptr, 0, M, N, K, dt, 0, 1, 0, stream, r);
}

extern "C" int mh_gemm(const void* A, int64_t lda, int a_kstrided, const void* B, int64_t ldb, int b_kstrided, void* C,
                       int64_t ldc, const void* bias, const void* resid, int64_t ldr, int M, int N, int K, int dt,
                       int epilogue, void* stream) {
  return gemm_impl(A, lda, a_kstrided, B, ldb, b_kstrided, C, ldc, bias, resid, ldr, M, N, K, dt, epilogue, 1, 0, stream);
}

extern "C" int mh_gemm_swiglu_fwd(const void* x, int64_t ldx, const void* Wgu, int64_t ldw, void* gu, int64_t ldgu, void* act,
                                  int64_t ldact, int M, int ff, int K, int dt, void* stream) {
  if (!gu || !act || ff <= 0 || (ff & 7) || (ldgu & 7) || (ldact & 7) || ((((uintptr_t)gu) | ((uintptr_t)act)) & 15u)) return MH_ERR_ARG;
  RopeSpec r;
  r.sw_mode = 1; r.sw_ff = ff; r.sw_out = act; r.sw_ldo = ldact;
  return gemm_impl(x, ldx, 0, Wgu, ldw, 0, gu, ldgu, nullptr, nullptr, 0, M, 2 * ff, K, dt, 0, 1, 0, stream, r);
}

extern "C" int mh_gemm_swiglu_bwd(const void* dy, int64_t lddy, const void* Wd, int64_t ldw, const void* gu, int64_t ldgu, void* dgu,
                                  int64_t lddgu, int M, int ff, int K, int dt, void* stream) {
  if (!gu || !dgu || ff <= 0 || (ff & 7) || (ldgu & 7) || (lddgu & 7) || ((((uintptr_t)gu) | ((uintptr_t)dgu)) & 15u)) return MH_ERR_ARG;
  RopeSpec r;
  r.sw_mode = 2; r.sw_ff = ff; r.sw_out = dgu; r.sw_ldo = lddgu; r.sw_in = gu; r.sw_ldi = ldgu;
  // dact[M, ff] = dy[M, K] * Wd[K, ff] (Wd is the down projection [d_model = K, ff] as stored: K-strided B); C stands in for
  // the staging checks only (dact is never written)
  return gemm_impl(dy, lddy, 0, Wd, ldw, 1, dgu, lddgu, nullptr, nullptr, 0, M, ff, K, dt, 0, 1, 0, stream, r);
}

// ---- fp8 operands (e4m3 bytes, one fp32 scale per row of A and per row of B), scaled-fp8 MFMA ------------------------
namespace {
// NCH > 0: the row (K <= 512 * NCH elements) is requested once, every 16-byte piece of a lane before any is consumed, and both passes
// (maximum, conversion) run on the packed registers; NCH = 0 walks the row twice with one dependent request per lane in flight.
template <int DT, int NCH>
__global__ __launch_bounds__(256) void quant_fp8_rows_k(const uint16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q,
                                                        int64_t ldq, float* __restrict__ sc, int R, int K) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const uint4* xr = (const uint4*)(x + (int64_t)row * ldx);
  const int nch = K >> 3;
  auto emit = [&](int c, const float* f, float inv) {
    int p0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, 0, false);
    p0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, p0, true);
    int p1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, 0, false);
    p1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, p1, true);
    *(uint2*)(q + (int64_t)row * ldq + c * 8) = make_uint2((unsigned)p0, (unsigned)p1);
  };
  float mx = 0.f;
  if constexpr (NCH > 0) {
    uint4 raw[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      raw[j] = xr[min(c, nch - 1)];
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      float f[8];
      unpack8<DT>(raw[j], f);  // (a clamped duplicate of the row's last piece changes no maximum)
#pragma unroll
      for (int i = 0; i < 8; ++i) mx = fmaxf(mx, fabsf(f[i]));
    }
    mx = wave_max(mx);
    const float s = mx > 0.f ? mx * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / s;
    if (lane == 0) sc[row] = s;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      if (c < nch) {
        float f[8];
        unpack8<DT>(raw[j], f);
        emit(c, f, inv);
      }
    }
  } else {
    for (int c = lane; c < nch; c += 64) {
      float f[8];
      unpack8<DT>(xr[c], f);
#pragma unroll
      for (int i = 0; i < 8; ++i) mx = fmaxf(mx, fabsf(f[i]));
    }
    mx = wave_max(mx);
    const float s = mx > 0.f ? mx * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / s;
    if (lane == 0) sc[row] = s;
    for (int c = lane; c < nch; c += 64) {
      float f[8];
      unpack8<DT>(xr[c], f);
      emit(c, f, inv);
    }
  }
}
}  // namespace

extern "C" int mh_quant_fp8_rows(const void* x, int64_t ldx, void* q, int64_t ldq, float* scales, int R, int K, int dt, void* stream) {
  if (!x || !q || !scales || R <= 0 || K <= 0 || (K & 7) || (ldx & 7) || (ldq & 7) || !aligned16(x) || (((uintptr_t)q) & 7u)) return MH_ERR_ARG;
  if (dt != MH_BF16 && dt != MH_F16) return MH_ERR_DTYPE;
  const dim3 grid((R + 3) / 4), block(256);
#define QR_GO(DT_, NCH_) \
  hipLaunchKernelGGL((quant_fp8_rows_k<DT_, NCH_>), grid, block, 0, as_stream(stream), (const uint16_t*)x, ldx, (uint8_t*)q, ldq, scales, R, K)
#define QR_DT(DT_) do { if (K <= 4096) QR_GO(DT_, 8); else if (K <= 12288) QR_GO(DT_, 24); else QR_GO(DT_, 0); } while (0)
  if (dt == MH_BF16) QR_DT(MH_BF16); else QR_DT(MH_F16);
#undef QR_DT
#undef QR_GO
  MH_LAUNCH_CHECK();
}

static int gemm_fp8_impl(const void* A8, int64_t lda, const float* sa, const void* B8, int64_t ldb, const float* sb, const void* b_exp, void* C,
                         int64_t ldc, const void* bias, const void* resid, int64_t ldr, int M, int N, int K, int dt_out,
                         int epilogue, const RopeSpec& fx, void* stream, unsigned* amax_ws = nullptr);

extern "C" int mh_gemm_fp8(const void* A8, int64_t lda, const float* sa, const void* B8, int64_t ldb, const float* sb, const void* b_exp,
                           void* C, int64_t ldc, const void* bias, const void* resid, int64_t ldr, int M, int N, int K, int dt_out,
                           int epilogue, void* stream) {
  return gemm_fp8_impl(A8, lda, sa, B8, ldb, sb, b_exp, C, ldc, bias, resid, ldr, M, N, K, dt_out, epilogue, RopeSpec(), stream);
}
// fp8 forms of mh_gemm_nt_rope / mh_gemm_swiglu_fwd (same staged store phases)
extern "C" int mh_gemm_fp8_rope(const void* A8, int64_t lda, const float* sa, const void* B8, int64_t ldb, const float* sb, const void* b_exp,
                                void* C, int64_t ldc, int M, int N, int K, int dt_out, const float* cos_sin, int S, int D, int rope_cols,
                                void* stream) {
  if (!cos_sin || S <= 0 || (D != 128 && D != 64) || rope_cols <= 0 || rope_cols > N || (rope_cols % D) || (N & 7) || (ldc & 7) ||
      ((((uintptr_t)C) & 15u) != 0))
    return MH_ERR_ARG;
  RopeSpec r;
  r.tab = cos_sin; r.S = S; r.D = D; r.cols = rope_cols;
  return gemm_fp8_impl(A8, lda, sa, B8, ldb, sb, b_exp, C, ldc, nullptr, nullptr, 0, M, N, K, dt_out, 0, r, stream);
}
extern "C" int mh_gemm_fp8_swiglu_fwd(const void* A8, int64_t lda, const float* sa, const void* B8, int64_t ldb, const float* sb,
                                      const void* b_exp, void* gu, int64_t ldgu, void* act, int64_t ldact, int M, int ff, int K, int dt_out,
                                      void* stream) {
  if (!gu || !act || ff <= 0 || (ff & 7) || (ldgu & 7) || (ldact & 7) || ((((uintptr_t)gu) | ((uintptr_t)act)) & 15u)) return MH_ERR_ARG;
  RopeSpec r;
  r.sw_mode = 1; r.sw_ff = ff; r.sw_out = act; r.sw_ldo = ldact;
  return gemm_fp8_impl(A8, lda, sa, B8, ldb, sb, b_exp, gu, ldgu, nullptr, nullptr, 0, M, 2 * ff, K, dt_out, 0, r, stream);
}

// fp8 form of mh_gemm_swiglu_bwd: dact[M, ff] = (sdy qdy)[M, K] (swt qwt)[ff, K]^T with qwt = rowquant(down_proj.weight^T)
// ([ff, d_model]: contraction over d_model), SwiGLU backward in the store phase (dact never reaches memory).
extern "C" int mh_gemm_fp8_swiglu_bwd(const void* dy8, int64_t lddy, const float* sdy, const void* WdT8, int64_t ldw, const float* swt,
                                      const void* wt_exp, const void* gu, int64_t ldgu, void* dgu, int64_t lddgu, int M, int ff, int K,
                                      int dt_out, void* stream) {
  if (!gu || !dgu || ff <= 0 || (ff & 7) || (ldgu & 7) || (lddgu & 7) || ((((uintptr_t)gu) | ((uintptr_t)dgu)) & 15u)) return MH_ERR_ARG;
  RopeSpec r;
  r.sw_mode = 2; r.sw_ff = ff; r.sw_out = dgu; r.sw_ldo = lddgu; r.sw_in = gu; r.sw_ldi = ldgu;
  return gemm_fp8_impl(dy8, lddy, sdy, WdT8, ldw, swt, wt_exp, dgu, lddgu, nullptr, nullptr, 0, M, ff, K, dt_out, 0, r, stream);
}
// The same with the maxima the quantisers behind it need ("produce row / column maxima in the kernels that write the tensors"): on return
// (stream order) amax_ws [M + 2 ff] holds the bit patterns of max |dgu| per row, then per column, of the tensor AS STORED - taken in the 4-wave kernel's
// store phase where that kernel runs, by one extra read of dgu (absmax_rc_k) where the 8-wave kernel does; mh_quant_fp8_rows_and_t_pre consumes it.
extern "C" int mh_gemm_fp8_swiglu_bwd_amax(const void* dy8, int64_t lddy, const float* sdy, const void* WdT8, int64_t ldw, const float* swt,
                                           const void* wt_exp, const void* gu, int64_t ldgu, void* dgu, int64_t lddgu, unsigned* amax_ws, int M, int ff,
                                           int K, int dt_out, void* stream) {
  if (!gu || !dgu || !amax_ws || ff <= 0 || (ff & 7) || (ldgu & 7) || (lddgu & 7) || ((((uintptr_t)gu) | ((uintptr_t)dgu)) & 15u)) return MH_ERR_ARG;
  if (hipMemsetAsync(amax_ws, 0, (size_t)(M + 2 * ff) * sizeof(unsigned), as_stream(stream)) != hipSuccess) return MH_ERR_ARG;
  RopeSpec r;
  r.sw_mode = 2; r.sw_ff = ff; r.sw_out = dgu; r.sw_ldo = lddgu; r.sw_in = gu; r.sw_ldi = ldgu;
  return gemm_fp8_impl(dy8, lddy, sdy, WdT8, ldw, swt, wt_exp, dgu, lddgu, nullptr, nullptr, 0, M, ff, K, dt_out, 0, r, stream, amax_ws);
}

static int gemm_fp8_impl(const void* A8, int64_t lda, const float* sa, const void* B8, int64_t ldb, const float* sb, const void* b_exp, void* C,
                         int64_t ldc, const void* bias, const void* resid, int64_t ldr, int M, int N, int K, int dt_out,
                         int epilogue, const RopeSpec& fx, void* stream, unsigned* amax_ws) {
  if (!A8 || !B8 || !sa || !sb || !C || M <= 0 || N <= 0 || K <= 0) return MH_ERR_ARG;
  if ((K % 128) || (lda & 15) || (ldb & 15) || !aligned16(A8) || !aligned16(B8)) return MH_ERR_ARG;
  if ((epilogue & MH_EPI_BIAS) && !bias) return MH_ERR_ARG;
  if ((epilogue & MH_EPI_RESIDUAL) && !resid) return MH_ERR_ARG;
  if (dt_out != MH_BF16 && dt_out != MH_F16) return MH_ERR_DTYPE;
  GemmArgs g;
  g.A = (const uint16_t*)A8; g.B = (const uint16_t*)B8; g.C = C;
  g.bias = (const uint16_t*)bias; g.resid = (const uint16_t*)resid;
  g.lda = lda / 2; g.ldb = ldb / 2; g.ldc = ldc; g.ldr = ldr;   // 2-byte units: a 128-byte LDS row = 128 k
  g.M = M; g.N = N; g.K = K / 2; g.epi = epilogue;
  const bool f32out = epilogue & MH_EPI_OUT_F32;
  g.vec_ok = (N % 4 == 0) && (ldc % 4 == 0) && ((((uintptr_t)C) & (f32out ? 15u : 7u)) == 0) &&
             (!(epilogue & MH_EPI_RESIDUAL) || ((ldr % 4 == 0) && ((((uintptr_t)resid) & 7u) == 0))) &&
             (!(epilogue & MH_EPI_BIAS) || ((((uintptr_t)bias) & 7u) == 0));
  g.splits = 1; g.c_split = 0; g.gm = g_gemm_gm;
  g.rope_tab = fx.tab; g.rope_S = fx.S; g.rope_D = fx.D; g.rope_cols = fx.cols;
  g.sw_mode = fx.sw_mode; g.sw_ff = fx.sw_ff; g.sw_out = fx.sw_out; g.sw_in = fx.sw_in; g.sw_ldo = fx.sw_ldo; g.sw_ldi = fx.sw_ldi;
  g.sc_m = sa; g.sc_n = sb;
  // b_exp: 16-byte header (int32 "any exponent non-zero" flag written by the quantiser + padding), then the exponent image
  g.sc_e = b_exp ? (const uint8_t*)b_exp + 16 : nullptr;
  g.sc_e_flag = (const int*)b_exp;
  g.sc_e_group = b_exp ? ((K / 128) * 64 + 4095) / 4096 * 4096 : 0;
  if (b_exp && (((uintptr_t)b_exp) & 15u)) return MH_ERR_ARG;
  {
    static void* zp = nullptr;
    if (!zp && hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero_row)) != hipSuccess) return MH_ERR_ARG;
    g.zero_row = (const uint16_t*)zp;
  }
  g.tiles_m = (M + 255) / 256;
  g.tiles_n = fx.sw_mode == 1 ? (fx.sw_ff + 127) / 128 : (N + 255) / 256;
  // 4-wave form (gemm_w4.hip) for exponent-free operands with a plain / residual / accumulating epilogue when the tiles fill the
  // chip; mh_gemm_force_kernel(4) = wherever it can run, (256) = never; auto: mh_gemm_w4_policy bit 3
  // (bit 3 of the policy mask: the plain / residual / accumulating products; bit 8: the fused forms - RoPE, SwiGLU forward / backward, the fp32 logits)
  g.amax_r = nullptr; g.amax_c = nullptr;
  if (g_force_kernel != 256 && w4_f8_can_run(g) &&
      (g_force_kernel == 4 || (g_force_kernel == 0 && (g_w4_mask & (w4_f8_is_fused(g) ? 256 : 8)) && (int64_t)g.tiles_m * g.tiles_n >= 192 && K >= 4096))) {
    if (amax_ws && fx.sw_mode == 2) { g.amax_r = amax_ws; g.amax_c = amax_ws + M; }
    return launch_gemm_w4_f8(g, dt_out, as_stream(stream));
  }
  const int rc = launch_gemm_nt_256_f8(g, dt_out, as_stream(stream));
  if (rc != MH_OK || !amax_ws || fx.sw_mode != 2) return rc;
  return launch_absmax_rc(fx.sw_out, fx.sw_ldo, amax_ws, amax_ws + M, M, 2 * fx.sw_ff, dt_out, as_stream(stream));
}

// quick-GELU in the GEMM store phase (CLIP MLP; HF CLIPMLP `fc1 -> quick_gelu -> fc2`): forward f1 = x W1^T + b1 AND a = quick_gelu(f1) from one
// launch; backward df1 = quick_gelu'(f1) * (dY W2) with dY W2 never stored.  Both work on the ROUNDED 16-bit tile: bit-identical to mh_gemm followed
// by mh_quick_gelu_fwd / _bwd.
extern "C" int mh_gemm_gelu_fwd(const void* x, int64_t ldx, const void* w1, int64_t ldw, const void* bias, void* f1, int64_t ldf, void* a, int64_t lda_out,
                                int M, int N, int K, int dt, void* stream) {
  if (!bias || !f1 || !a || (N & 7) || (ldf & 7) || (lda_out & 7) || ((((uintptr_t)f1) | ((uintptr_t)a)) & 15u) || (((uintptr_t)bias) & 7u)) return MH_ERR_ARG;
  RopeSpec r;
  r.sw_mode = 3; r.sw_out = a; r.sw_ldo = lda_out;
  return gemm_impl(x, ldx, 0, w1, ldw, 0, f1, ldf, bias, nullptr, 0, M, N, K, dt, MH_EPI_BIAS, 1, 0, stream, r);
}
extern "C" int mh_gemm_gelu_bwd(const void* dy, int64_t lddy, const void* w2, int64_t ldw, const void* f1, int64_t ldf, void* df1, int64_t lddf,
                                int M, int N, int K, int dt, void* stream) {
  // dy [M, K = d], w2 = fc2.weight [d, N = ff] read K-strided, f1 / df1 [M, N]
  if (!f1 || !df1 || (N & 7) || (ldf & 7) || (lddf & 7) || ((((uintptr_t)f1) | ((uintptr_t)df1)) & 15u)) return MH_ERR_ARG;
  RopeSpec r;
  r.sw_mode = 4; r.sw_out = df1; r.sw_ldo = lddf; r.sw_in = f1; r.sw_ldi = ldf;
  return gemm_impl(dy, lddy, 0, w2, ldw, 1, df1, lddf, nullptr, nullptr, 0, M, N, K, dt, 0, 1, 0, stream, r);
}

extern "C" int mh_wgrad_grouped(const MhWgradProblem* pr, int n, int T, int dt, void* stream) {
  if (!pr || n < 1 || n > 8 || T <= 0) return MH_ERR_ARG;
  if (dt != MH_BF16 && dt != MH_F16) return MH_ERR_DTYPE;
  GemmArgs gs[8];
  for (int i = 0; i < n; ++i) {
    const MhWgradProblem& q = pr[i];
    if (!q.dy || !q.x || !q.out || q.M <= 0 || q.N <= 0) return MH_ERR_ARG;
    if ((q.lddy & 7) || (q.ldx & 7) || !aligned16(q.dy) || !aligned16(q.x)) return MH_ERR_SHAPE;
    GemmArgs g{};
    g.A = (const uint16_t*)q.dy; g.B = (const uint16_t*)q.x; g.C = q.out;
    g.lda = q.lddy; g.ldb = q.ldx; g.ldc = q.ldo;
    g.M = q.M; g.N = q.N; g.K = T; g.epi = q.accumulate ? MH_EPI_ACCUM : 0;
    g.splits = 1; g.gm = g_gemm_gm; g.vec_ok = 1;
    g.tiles_m = (q.M + 255) / 256; g.tiles_n = (q.N + 255) / 256;
    if (!w4_can_run(g, 1, 1)) return MH_ERR_SHAPE;
    gs[i] = g;
  }
  return launch_gemm_w4_grouped(gs, n, dt, g_gemm_gm ? g_gemm_gm : 4, as_stream(stream));
}

extern "C" int mh_gemm_splitk_max(int M, int N, int K) {
  const int64_t tiles = (int64_t)((M + 255) / 256) * ((N + 255) / 256);
  const int nk = (K + BK - 1) / BK;
  if (tiles >= 128 || nk < 16) return 1;
  int s = (int)(256 / tiles);  // one round of blocks: tiles * s <= 256 CUs
  if (s > 16) s = 16;
  while (s > 1 && (nk + s - 1) / s < 8) --s;           // >= 8 K-tiles per split
  while (s > 1 && (s - 1) * ((nk + s - 1) / s) >= nk) --s;  // no empty split
  return s;
}

extern "C" int mh_gemm_splitk(const void* A, int64_t lda, int a_kstrided, const void* B, int64_t ldb, int b_kstrided, void* C,
                              int64_t ldc, int M, int N, int K, int dt, int accumulate, int out_f32, int splits, float* ws,
                              void* stream) {
  if (splits < 1 || splits > 64) return MH_ERR_ARG;
  if (splits == 1)
    return gemm_impl(A, lda, a_kstrided, B, ldb, b_kstrided, C, ldc, nullptr, nullptr, 0, M, N, K, dt,
                     (accumulate ? MH_EPI_ACCUM : 0) | (out_f32 ? MH_EPI_OUT_F32 : 0), 1, 0, stream);
  if (!ws || (N & 3) || (ldc & 3)) return MH_ERR_ARG;
  const int nk = (K + BK - 1) / BK;
  if ((int64_t)(splits - 1) * ((nk + splits - 1) / splits) >= nk) return MH_ERR_ARG;  // an empty split
  const int rc = gemm_impl(A, lda, a_kstrided, B, ldb, b_kstrided, ws, N, nullptr, nullptr, 0, M, N, K, dt, MH_EPI_OUT_F32, splits,
                           (int64_t)M * N, stream);
  if (rc != MH_OK) return rc;
  const int64_t quads = (int64_t)M * (N >> 2);
  const unsigned grid = (unsigned)((quads + 255) / 256);
  if (dt == MH_F16)
    hipLaunchKernelGGL(splitk_reduce_k<MH_F16>, dim3(grid), dim3(256), 0, as_stream(stream), ws, splits, (int64_t)M * N, C, ldc, M, N, out_f32, accumulate);
  else
    hipLaunchKernelGGL(splitk_reduce_k<MH_BF16>, dim3(grid), dim3(256), 0, as_stream(stream), ws, splits, (int64_t)M * N, C, ldc, M, N, out_f32, accumulate);
  MH_LAUNCH_CHECK();
}

// Split-K with the full epilogue: `splits` blocks per output tile write fp32 partials to ws (splits * M * N floats), one fixed-order
// pass sums them and applies bias / quick-GELU / residual / accumulate / the fp32 store exactly as the one-pass kernel's store phase does.
// For products with few output tiles and a long contraction (a 613-token sequence gives the o / down projections and every dgrad
// 48 tiles for 256 CUs).  splits from mh_gemm_splitk_max; 1 = the plain call.
extern "C" int mh_gemm_splitk_epi(const void* A, int64_t lda, int a_kstrided, const void* B, int64_t ldb, int b_kstrided, void* C, int64_t ldc,
                                  const void* bias, const void* resid, int64_t ldr, int M, int N, int K, int dt, int epilogue, int splits,
                                  float* ws, void* stream) {
  if (splits < 1 || splits > 64) return MH_ERR_ARG;
  if (splits == 1) return gemm_impl(A, lda, a_kstrided, B, ldb, b_kstrided, C, ldc, bias, resid, ldr, M, N, K, dt, epilogue, 1, 0, stream);
  if (!ws || !C || (N & 3) || (ldc & 3)) return MH_ERR_ARG;
  if ((epilogue & MH_EPI_BIAS) && !bias) return MH_ERR_ARG;
  if ((epilogue & MH_EPI_RESIDUAL) && !resid) return MH_ERR_ARG;
  if (dt != MH_BF16 && dt != MH_F16) return MH_ERR_DTYPE;
  const int nk = (K + BK - 1) / BK;
  if ((int64_t)(splits - 1) * ((nk + splits - 1) / splits) >= nk) return MH_ERR_ARG;  // an empty split
  const int rc = gemm_impl(A, lda, a_kstrided, B, ldb, b_kstrided, ws, N, nullptr, nullptr, 0, M, N, K, dt, MH_EPI_OUT_F32, splits,
                           (int64_t)M * N, stream);
  if (rc != MH_OK) return rc;
  GemmArgs g{};
  g.C = C; g.bias = (const uint16_t*)bias; g.resid = (const uint16_t*)resid;
  g.ldc = ldc; g.ldr = ldr; g.M = M; g.N = N; g.K = K; g.epi = epilogue;
  const bool f32out = epilogue & MH_EPI_OUT_F32;
  g.vec_ok = (ldc % 4 == 0) && ((((uintptr_t)C) & (f32out ? 15u : 7u)) == 0) &&
             (!(epilogue & MH_EPI_RESIDUAL) || ((ldr % 4 == 0) && ((((uintptr_t)resid) & 7u) == 0))) &&
             (!(epilogue & MH_EPI_BIAS) || ((((uintptr_t)bias) & 7u) == 0));
  const int64_t quads = (int64_t)M * (N >> 2);
  const unsigned grid = (unsigned)((quads + 255) / 256);
  if (dt == MH_F16)
    hipLaunchKernelGGL(splitk_reduce_epi_k<MH_F16>, dim3(grid), dim3(256), 0, as_stream(stream), ws, splits, (int64_t)M * N, g);
  else
    hipLaunchKernelGGL(splitk_reduce_epi_k<MH_BF16>, dim3(grid), dim3(256), 0, as_stream(stream), ws, splits, (int64_t)M * N, g);
  MH_LAUNCH_CHECK();
}

static int gemm_impl(const void* A, int64_t lda, int a_kstrided, const void* B, int64_t ldb, int b_kstrided, void* C,
                     int64_t ldc, const void* bias, const void* resid, int64_t ldr, int M, int N, int K, int dt,
                     int epilogue, int splits, int64_t c_split, void* stream, RopeSpec rope) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return MH_ERR_ARG;
  const bool both_ks = a_kstrided && b_kstrided;  // the only form whose K need not be a multiple of 64 (zero rows)
  if ((!both_ks && K % BK != 0) || (lda & 7) || (ldb & 7) || !aligned16(A) || !aligned16(B)) return MH_ERR_ARG;
  if ((a_kstrided && (M & 7)) || (b_kstrided && (N & 7))) return MH_ERR_ARG;
  if ((epilogue & MH_EPI_BIAS) && !bias) return MH_ERR_ARG;
  if ((epilogue & MH_EPI_RESIDUAL) && !resid) return MH_ERR_ARG;
  if (dt != MH_BF16 && dt != MH_F16) return MH_ERR_DTYPE;
  GemmArgs g;
  g.A = (const uint16_t*)A; g.B = (const uint16_t*)B; g.C = C;
  g.bias = (const uint16_t*)bias; g.resid = (const uint16_t*)resid;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr;
  g.M = M; g.N = N; g.K = K; g.epi = epilogue;
  g.splits = splits; g.c_split = c_split; g.gm = g_gemm_gm;
  g.sc_m = nullptr; g.sc_n = nullptr; g.sc_e = nullptr; g.sc_e_flag = nullptr; g.sc_e_group = 0; g.amax_r = nullptr; g.amax_c = nullptr;
  g.rope_tab = rope.tab; g.rope_S = rope.S; g.rope_D = rope.D; g.rope_cols = rope.cols;
  g.sw_mode = rope.sw_mode; g.sw_ff = rope.sw_ff; g.sw_out = rope.sw_out; g.sw_in = rope.sw_in; g.sw_ldo = rope.sw_ldo; g.sw_ldi = rope.sw_ldi;
  {
    static void* zp = nullptr;
    if (!zp && hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero_row)) != hipSuccess) return MH_ERR_ARG;
    g.zero_row = (const uint16_t*)zp;
  }
  const bool f32out = epilogue & MH_EPI_OUT_F32;
  g.vec_ok = (N % 4 == 0) && (ldc % 4 == 0) && ((((uintptr_t)C) & (f32out ? 15u : 7u)) == 0) &&
             (!(epilogue & MH_EPI_RESIDUAL) || ((ldr % 4 == 0) && ((((uintptr_t)resid) & 7u) == 0))) &&
             (!(epilogue & MH_EPI_BIAS) || ((((uintptr_t)bias) & 7u) == 0));
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_nt_128<MH_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE128);
    hipFuncSetAttribute((const void*)gemm_nt_128<MH_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE128);
    attr_set = true;
  }
  // 256^2 tiles when they fill the chip (>= ~1 block per CU); 128^2 otherwise
  const int64_t t256 = (int64_t)((M + 255) / 256) * ((N + 255) / 256);
  bool big = t256 >= 192;
  if (g_force_kernel == 128) big = false;
  if (g_force_kernel == 256) big = true;
#ifdef MH_DEV_ARMS  // alternative tilings kept for A/B timing (tools/dev_arms/): compiled only into the dev library
  if ((g_force_kernel == 32 || g_force_kernel == 33) && !a_kstrided && !b_kstrided && splits == 1 && !rope.tab && !rope.sw_mode) {  // A/B arms: MFMA 32x32x16 fragments (33: two phases per K-tile)
    g_m32_kcut = g_force_kernel == 33;
    g.tiles_m = (M + 255) / 256;
    g.tiles_n = (N + 255) / 256;
    return launch_gemm_nt_256_m32(g, dt, as_stream(stream));
  }
  if (g_force_kernel >= 40 && g_force_kernel <= 48 && !a_kstrided && !b_kstrided && splits == 1 && !rope.tab && !rope.sw_mode &&
      (int64_t)M * lda * 2 < (1ll << 31) && (int64_t)N * ldb * 2 < (1ll << 31)) {  // A/B arm: four waves x 128x128 (gemm256w4.hip)
    const int e = epilogue;
    const bool known = e == 0 || e == MH_EPI_RESIDUAL || e == MH_EPI_BIAS || e == (MH_EPI_BIAS | MH_EPI_QUICK_GELU) || e == (MH_EPI_BIAS | MH_EPI_RESIDUAL);
    if (known && g.vec_ok && (N % 8 == 0) && (ldc % 8 == 0) && ((((uintptr_t)C) & 15u) == 0)) {  // staged epilogue only
      g.tiles_m = (M + 255) / 256;
      g.tiles_n = (N + 255) / 256;
      return launch_gemm_nt_w4(g, dt, as_stream(stream), g_force_kernel - 40);
    }
  }
  if (g_force_kernel == 88 && !a_kstrided && !b_kstrided && splits == 1 && !rope.tab && !rope.sw_mode &&
      (int64_t)M * lda * 2 < (1ll << 31) && (int64_t)N * ldb * 2 < (1ll << 31)) {  // A/B arm: 8 waves, dense stream (gemm256w8.hip)
    const int e = epilogue;
    const bool known = e == 0 || e == MH_EPI_RESIDUAL || e == MH_EPI_BIAS || e == (MH_EPI_BIAS | MH_EPI_QUICK_GELU) || e == (MH_EPI_BIAS | MH_EPI_RESIDUAL);
    if (known && g.vec_ok && (N % 8 == 0) && (ldc % 8 == 0) && ((((uintptr_t)C) & 15u) == 0)) {
      g.tiles_m = (M + 255) / 256;
      g.tiles_n = (N + 255) / 256;
      return launch_gemm_nt_w8(g, dt, as_stream(stream));
    }
  }
#endif
  // The 4-wave kernel (gemm_w4.hip: 128x128 per wave, a third fewer LDS bytes per flop) where it measures faster than the 8-wave
  // one (profiles/r03_gemm_w4_ab.txt): K-strided weight gradients with a long token contraction.  mh_gemm_force_kernel(4) sends
  // everything it can run there, (256) nothing.
  {
    g.tiles_m = (M + 255) / 256;
    g.tiles_n = rope.sw_mode == 1 ? (rope.sw_ff + 127) / 128 : (N + 255) / 256;  // (SwiGLU forward: a tile = 2 x (64 gate + 64 up) columns)
    const bool can = g_force_kernel != 256 && g_force_kernel != 128 && w4_can_run(g, a_kstrided, b_kstrided) && w4_has_kernel(g, a_kstrided, b_kstrided);
    // bit 7 of the policy mask: the fused forward forms go to the 4-wave kernel at ANY size (a 613-token sequence has 144 tiles) - they
    // rotate / gate the fp32 accumulators there, one rounding less than the 8-wave kernel's staged forms (parity; measured: DESIGN §4)
    const bool fused_any = (g_w4_mask & 128) && (rope.tab || rope.sw_mode == 1);
    const bool fills = big || fused_any || (int64_t)g.tiles_m * g.tiles_n * splits >= 192;
    const bool want = g_force_kernel == 4 || (g_force_kernel == 0 && fills && w4_policy(a_kstrided, b_kstrided, M, N, K, epilogue, rope));
    if (can && want) {
      if (g_w4_half && w4_has_half(g, a_kstrided, b_kstrided) && (g_w4_half == 2 || w4_half_pays(M, g.tiles_n, splits > 1 ? splits : 1))) {
        g.tiles_m = (M + 127) / 128;
        return launch_gemm_w4(g, dt, a_kstrided, b_kstrided, as_stream(stream), 1);
      }
      return launch_gemm_w4(g, dt, a_kstrided, b_kstrided, as_stream(stream), 0);
    }
  }
  if (rope.sw_mode == 1) {  // a tile = 128 gate + 128 up columns
    g.tiles_m = (M + 255) / 256;
    g.tiles_n = (rope.sw_ff + 127) / 128;
    return launch_gemm_256(g, dt, 0, 0, as_stream(stream));
  }
  if (big || a_kstrided || b_kstrided || splits > 1 || rope.tab || rope.sw_mode) {  // K-strided operands / split-K / fused RoPE, SwiGLU exist only in the 8-wave 256-tile kernel
    g.tiles_m = (M + 255) / 256;
    g.tiles_n = (N + 255) / 256;
    return launch_gemm_256(g, dt, a_kstrided, b_kstrided, as_stream(stream));
  } else {
    g.tiles_m = (M + BM128 - 1) / BM128;
    g.tiles_n = (N + BN128 - 1) / BN128;
    const int grid = g.tiles_m * g.tiles_n;
    if (dt == MH_BF16)
      hipLaunchKernelGGL(gemm_nt_128<MH_BF16>, dim3(grid), dim3(256), 2 * STAGE128, as_stream(stream), g);
    else
      hipLaunchKernelGGL(gemm_nt_128<MH_F16>, dim3(grid), dim3(256), 2 * STAGE128, as_stream(stream), g);
  }
  MH_LAUNCH_CHECK();
}

extern "C" int mh_transpose16(const void* in, int64_t ldi, void* out, int64_t ldo, int R, int C, int R_pad, void* stream) {
  if (!in || !out || R <= 0 || C <= 0 || R_pad < R || ldo < R_pad) return MH_ERR_ARG;
  dim3 grid((C + 63) / 64, (R_pad + 63) / 64);
  hipLaunchKernelGGL(transpose16_kernel, grid, dim3(256), 0, as_stream(stream), (const uint16_t*)in, ldi,
                     (uint16_t*)out, ldo, R, C, R_pad);
  MH_LAUNCH_CHECK();
}
