// bf16/f16 MFMA GEMM, C[M,N] = A[M,K] * B[N,K]^T with fused epilogues (gfx950).
//
// v1 structure ("2-phase", one barrier per K-tile):
//   * 128x128x64 block tile, 256 threads = 4 waves in 2x2, each wave a 64x64 output tile made of
//     4x4 MFMA 16x16x32 fragments (64 fp32 accumulators / lane).
//   * both operands are K-contiguous ("NT"): a tile row is 64 k = 128 B = 8 x 16-B chunks.
//   * global -> LDS by `global_load_lds_dwordx4` (no VGPR round trip), double buffered: the loads
//     of tile t+1 are in flight while tile t is multiplied; a single s_waitcnt vmcnt(0)+barrier
//     per K-tile.
//   * LDS image is lane-linear (a hard requirement of global_load_lds), so the bank swizzle is
//     applied on the per-lane GLOBAL source address and again on the ds_read address
//     (cdna_hip_programming.md rule 21): chunk' = chunk ^ ((row >> 1) & 7).  With 128-B rows two
//     rows share one 256-B bank row, so a ds_read_b128 lane group (16 lanes, rows r..r+15 at one
//     logical chunk) lands on 16 distinct 16-B slots: conflict-free.
//   * operands are passed to the MFMA swapped (B-fragment as the A operand) so that each lane ends
//     up with 4 CONSECUTIVE n for one m: the epilogue reads bias/residual and writes C as 8-byte
//     (16-bit C) or 16-byte (fp32 C) vectors.
//   * block id -> tile: XCD-aware remap (block b runs on XCD b%8; each XCD gets a contiguous tile
//     range, bijective for any grid) + grouped ordering (8 m-tiles per group) so co-resident
//     blocks of one XCD share A/B panels in that XCD's private L2.
// M and N edges: loads clamp the row index, stores are predicated.  K % 64 == 0 is required.
#include "mh_common.h"

namespace {

struct GemmArgs {
  const uint16_t* A;
  const uint16_t* B;
  void* C;
  const uint16_t* bias;
  const uint16_t* resid;
  int64_t lda, ldb, ldc, ldr;
  int M, N, K, epi, tiles_m, tiles_n, vec_ok;
};

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 32 KiB
constexpr int A_BYTES = BM * BK * 2;             // 16 KiB

template <int DT>
__device__ __forceinline__ float epi_apply(float v, int epi) {
  if (epi & MH_EPI_QUICK_GELU) v = v / (1.0f + __expf(-1.702f * v));
  return v;
}

template <int DT>
__global__ __launch_bounds__(256, 2) void gemm_nt_128(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- block -> tile (XCD-aware, grouped) ----
  int tile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tm, tn;
  {
    constexpr int GM = 8;
    const int per_group = GM * g.tiles_n;
    const int group = tile / per_group;
    const int first_m = group * GM;
    const int gsize = min(g.tiles_m - first_m, GM);
    const int in_g = tile - group * per_group;
    tm = first_m + in_g % gsize;
    tn = in_g / gsize;
  }
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- per-thread staging sources: 4 chunks of A and 4 of B per K-tile ----
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
    char* sA = smem + s * STAGE_BYTES;
    char* sB = sA + A_BYTES;
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
    const char* sA = smem + (kt & 1) * STAGE_BYTES;
    const char* sB = sA + A_BYTES;
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

  // ---- epilogue: lane holds C[m = .. + (lane&15)][n = .. + 4*(lane>>4) + r], r = 0..3 ----
  const int epi = g.epi;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + (lane & 15);
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
      if (n >= g.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (g.vec_ok) {
        if (epi & MH_EPI_BIAS) {
          const uint2 bb = *(const uint2*)(g.bias + n);
          float b0, b1, b2, b3;
          unpack2<DT>(bb.x, b0, b1);
          unpack2<DT>(bb.y, b2, b3);
          v[0] += b0; v[1] += b1; v[2] += b2; v[3] += b3;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = epi_apply<DT>(v[r], epi);
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
          x = epi_apply<DT>(x, epi);
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
  }
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

}  // namespace

extern "C" int mh_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                          const void* bias, const void* resid, int64_t ldr, int M, int N, int K, int dt,
                          int epilogue, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return MH_ERR_ARG;
  if (K % BK != 0 || (lda & 7) || (ldb & 7) || !aligned16(A) || !aligned16(B)) return MH_ERR_ARG;
  if ((epilogue & MH_EPI_BIAS) && !bias) return MH_ERR_ARG;
  if ((epilogue & MH_EPI_RESIDUAL) && !resid) return MH_ERR_ARG;
  if (dt != MH_BF16 && dt != MH_F16) return MH_ERR_DTYPE;
  GemmArgs g;
  g.A = (const uint16_t*)A; g.B = (const uint16_t*)B; g.C = C;
  g.bias = (const uint16_t*)bias; g.resid = (const uint16_t*)resid;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr;
  g.M = M; g.N = N; g.K = K; g.epi = epilogue;
  g.tiles_m = (M + BM - 1) / BM;
  g.tiles_n = (N + BN - 1) / BN;
  const bool f32out = epilogue & MH_EPI_OUT_F32;
  g.vec_ok = (N % 4 == 0) && (ldc % 4 == 0) && ((((uintptr_t)C) & (f32out ? 15u : 7u)) == 0) &&
             (!(epilogue & MH_EPI_RESIDUAL) || ((ldr % 4 == 0) && ((((uintptr_t)resid) & 7u) == 0))) &&
             (!(epilogue & MH_EPI_BIAS) || ((((uintptr_t)bias) & 7u) == 0));
  const int grid = g.tiles_m * g.tiles_n;
  const size_t lds = 2 * STAGE_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_nt_128<MH_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)gemm_nt_128<MH_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  if (dt == MH_BF16)
    hipLaunchKernelGGL(gemm_nt_128<MH_BF16>, dim3(grid), dim3(256), lds, as_stream(stream), g);
  else
    hipLaunchKernelGGL(gemm_nt_128<MH_F16>, dim3(grid), dim3(256), lds, as_stream(stream), g);
  MH_LAUNCH_CHECK();
}

extern "C" int mh_transpose16(const void* in, int64_t ldi, void* out, int64_t ldo, int R, int C, int R_pad, void* stream) {
  if (!in || !out || R <= 0 || C <= 0 || R_pad < R || ldo < R_pad) return MH_ERR_ARG;
  dim3 grid((C + 63) / 64, (R_pad + 63) / 64);
  hipLaunchKernelGGL(transpose16_kernel, grid, dim3(256), 0, as_stream(stream), (const uint16_t*)in, ldi,
                     (uint16_t*)out, ldo, R, C, R_pad);
  MH_LAUNCH_CHECK();
}
