// Flash attention backward for gfx950: dQ, dK, dV from Q, K, V, O, dO, LSE without materialising P.
//
// Replaces flash-attn's backward behind flash_attn_varlen_qkvpacked_func
// (mmgpt/utils/llama_flash_attn_monkey_patch.py:82,93; autograd of those calls).
//
// Two deterministic kernels (no atomics), both built on MFMA 32x32x16 with the same idea as the
// forward: orient every product so that the accumulator of the score-like matrix can be fed
// straight back as the B operand of the next product (its row index lives in registers, which is
// exactly the contraction index), and bake the implied contraction-order permutation (bits 2<->3
// inside each group of 16) into pre-transposed operand copies made by mh_attn_prep_v's kernel:
//
//  A) dK/dV, one block per 128 keys (4 waves x 32 keys, K and V fragments live in registers),
//     streaming 32-query tiles of {Q, dO, Q^T, dO^T} through LDS (global_load_lds, double buffer):
//        S  [q,kv] = Q K^T          A = Q  (LDS rows)    B = K  (regs)
//        dP [q,kv] = dO V^T         A = dO (LDS rows)    B = V  (regs)
//        P = exp2(S*c - lse[q]*log2e),  dS = P o (dP - delta[q]) * scale      (fp32, in registers)
//        dV^T[d,kv] += dO^T P       A = dO^T (LDS rows)  B = P  (accumulator regs -> bf16x8)
//        dK^T[d,kv] += Q^T  dS      A = Q^T  (LDS rows)  B = dS (accumulator regs)
//  B) dQ, one block per 128 queries (4 waves x 32 queries, Q and dO fragments in registers),
//     streaming 32-key tiles of {K, V, K^T}:
//        S^T [kv,q] = K Q^T,  dP^T[kv,q] = V dO^T,   dQ^T[d,q] += K^T dS^T
//  delta[q] = rowsum(dO o O) is produced by a small HBM-bound kernel first.
// lse and delta are [B, H, S_pad] fp32 (S_pad = round_up(S, 64)) so per-lane float4 loads stay aligned.
#include "mh_common.h"

namespace {

struct BwdArgs {
  const uint16_t *q, *k, *v, *o, *dout;
  const uint16_t *qt, *dot, *kt;  // [B, H, D, S_pad] permuted transposes
  const float* lse;
  float* delta;
  uint16_t *dq, *dk, *dv;
  const int32_t* seqlens;
  int64_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  int B, S, H, S_pad;
  float scale, scale_log2;
};

template <int D> struct RSwz;  // swizzle of a [rows][D] 16-bit tile (row = D*2 bytes)
template <> struct RSwz<128> { static __device__ __forceinline__ int f(int row) { return row & 15; } };
template <> struct RSwz<64> { static __device__ __forceinline__ int f(int row) { return (row >> 1) & 7; } };
// swizzle of a [rows][32] 16-bit tile (64-byte rows, 4 chunks): 4 rows share a 256-B bank row
__device__ __forceinline__ int tswz(int row) { return (row >> 2) & 3; }

template <int DT, int D>
__global__ __launch_bounds__(256) void delta_k(BwdArgs a) {
  // one wave per (token, head); lanes stride the head dim 2 (D=128) or 1 (D=64) elements at a time
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t idx = (int64_t)blockIdx.x * 4 + wave;  // over B*S*H
  const int64_t total = (int64_t)a.B * a.S * a.H;
  if (idx >= total) return;
  const int h = (int)(idx % a.H);
  const int64_t t = idx / a.H;
  const int b = (int)(t / a.S), s = (int)(t % a.S);
  float acc = 0.f;
  if constexpr (D == 128) {
    const uint32_t x = *(const uint32_t*)(a.o + t * a.ldo + (int64_t)h * D + lane * 2);
    const uint32_t y = *(const uint32_t*)(a.dout + t * a.lddo + (int64_t)h * D + lane * 2);
    float x0, x1, y0, y1;
    unpack2<DT>(x, x0, x1);
    unpack2<DT>(y, y0, y1);
    acc = x0 * y0 + x1 * y1;
  } else {
    acc = ld16<DT>(a.o[t * a.ldo + (int64_t)h * D + lane]) * ld16<DT>(a.dout[t * a.lddo + (int64_t)h * D + lane]);
  }
  acc = wave_sum(acc);
  if (lane == 0) a.delta[((int64_t)b * a.H + h) * a.S_pad + s] = acc;
}

// ------------------------------------------------------------------------------------------------
// Kernel A: dK, dV
// ------------------------------------------------------------------------------------------------
template <int DT, int D, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_k(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CPR = D / 8;                 // chunks per row-major row
  constexpr int TILE = 32 * D * 2;           // bytes of one [32][D] (or [D][32]) tile
  constexpr int STAGE = 4 * TILE;            // Q, dO, Q^T, dO^T
  constexpr int KSTEPS = D / 16, DBLK = D / 32;
  constexpr int NLD = TILE / (256 * 16);     // glds per thread per tile kind (2 for D=128, 1 for D=64)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int S = a.S;
  const int len = a.seqlens ? min(a.seqlens[b], S) : S;
  const int kv0 = blockIdx.x * 128;
  const int kvrow = kv0 + wave * 32 + l31;
  uint16_t* dkp = a.dk + ((int64_t)b * S + kvrow) * a.lddk + (int64_t)h * D;
  uint16_t* dvp = a.dv + ((int64_t)b * S + kvrow) * a.lddv + (int64_t)h * D;

  if (kv0 >= len) {
    if (kvrow < S) {
      for (int d = hi * (D / 2); d < (hi + 1) * (D / 2); d += 4) {
        *(uint2*)(dkp + d) = make_uint2(0, 0);
        *(uint2*)(dvp + d) = make_uint2(0, 0);
      }
    }
    return;
  }

  // K, V fragments (B operands): lane holds X[kvrow][16*ks + 8*hi .. +8]
  uint4 kf[KSTEPS], vf[KSTEPS];
  {
    const int kr = min(kvrow, S - 1);
    const uint16_t* kp = a.k + ((int64_t)b * S + kr) * a.ldk + (int64_t)h * D + 8 * hi;
    const uint16_t* vp = a.v + ((int64_t)b * S + kr) * a.ldv + (int64_t)h * D + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      kf[ks] = *(const uint4*)(kp + 16 * ks);
      vf[ks] = *(const uint4*)(vp + 16 * ks);
    }
  }

  const int q_begin = CAUSAL ? kv0 : 0;
  const int ntiles = (len - q_begin + 31) / 32;

  // staging sources
  int rrow[NLD], rcol[NLD], trow[NLD], tcol[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int qd = i * 256 + tid;
    rrow[i] = qd / CPR;
    rcol[i] = ((qd % CPR) ^ RSwz<D>::f(rrow[i])) * 8;
    trow[i] = qd >> 2;  // d
    tcol[i] = ((qd & 3) ^ tswz(trow[i])) * 8;
  }
  const int64_t bh_t = ((int64_t)b * a.H + h) * D;
  auto stage = [&](int s, int q0) {
    char* base = smem + s * STAGE;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int qr = min(q0 + rrow[i], S - 1);
      const int64_t roff = ((int64_t)b * S + qr);
      glds16(a.q + roff * a.ldq + (int64_t)h * D + rcol[i], base + 0 * TILE + (i * 256 + wave * 64) * 16);
      glds16(a.dout + roff * a.lddo + (int64_t)h * D + rcol[i], base + 1 * TILE + (i * 256 + wave * 64) * 16);
      glds16(a.qt + (bh_t + trow[i]) * a.S_pad + q0 + tcol[i], base + 2 * TILE + (i * 256 + wave * 64) * 16);
      glds16(a.dot + (bh_t + trow[i]) * a.S_pad + q0 + tcol[i], base + 3 * TILE + (i * 256 + wave * 64) * 16);
    }
  };

  f32x16_t dvacc[DBLK], dkacc[DBLK];
#pragma unroll
  for (int i = 0; i < DBLK; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dvacc[i][r] = 0.f; dkacc[i][r] = 0.f; }

  const int r_off = l31 * (D * 2);
  const int r_swz = RSwz<D>::f(l31);
  const float* lse_row = a.lse + ((int64_t)b * a.H + h) * a.S_pad;
  const float* dl_row = a.delta + ((int64_t)b * a.H + h) * a.S_pad;
  const float sc = a.scale_log2;

  stage(0, q_begin);
  for (int j = 0; j < ntiles; ++j) {
    const int q0 = q_begin + j * 32;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (j + 1 < ntiles) stage((j + 1) & 1, q0 + 32);
    // causal: this wave's keys all above every query of the tile -> nothing to do
    if (CAUSAL && (kv0 + wave * 32 > q0 + 31)) continue;
    const char* sQ = smem + (j & 1) * STAGE;
    const char* sDO = sQ + TILE;
    const char* sQT = sQ + 2 * TILE;
    const char* sDOT = sQ + 3 * TILE;

    f32x16_t sacc, pacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int coff = ((2 * ks + hi) ^ r_swz) << 4;
      const uint4 qa = *(const uint4*)(sQ + r_off + coff);
      const uint4 da = *(const uint4*)(sDO + r_off + coff);
      sacc = mfma32<DT>(qa, kf[ks], sacc);
      pacc = mfma32<DT>(da, vf[ks], pacc);
    }
    // P and dS (rows = queries live in registers: q = q0 + (r&3) + 8*(r>>2) + 4*hi)
    float pv[16], dsv[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int qb = q0 + 8 * g + 4 * hi;
      const float4 l4 = *(const float4*)(lse_row + qb);
      const float4 d4 = *(const float4*)(dl_row + qb);
      const float ls[4] = {l4.x, l4.y, l4.z, l4.w};
      const float dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        const int q = qb + e;
        const bool ok = (q < len) && (kvrow < len) && (!CAUSAL || kvrow <= q);
        const float p = ok ? exp2f(sacc[r] * sc - ls[e] * 1.4426950408889634f) : 0.f;
        pv[r] = p;
        dsv[r] = ok ? p * (pacc[r] - dl[e]) * a.scale : 0.f;
      }
    }
    uint4 pf[2], dsf[2];
    pf[0] = pack8<DT>(pv); pf[1] = pack8<DT>(pv + 8);
    dsf[0] = pack8<DT>(dsv); dsf[1] = pack8<DT>(dsv + 8);
#pragma unroll
    for (int i = 0; i < DBLK; ++i) {
      const int row = 32 * i + l31;
      const int tsw = tswz(row);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int off = row * 64 + (((2 * s + hi) ^ tsw) << 4);
        const uint4 dot_a = *(const uint4*)(sDOT + off);
        const uint4 qt_a = *(const uint4*)(sQT + off);
        dvacc[i] = mfma32<DT>(dot_a, pf[s], dvacc[i]);
        dkacc[i] = mfma32<DT>(qt_a, dsf[s], dkacc[i]);
      }
    }
  }

  if (kvrow < S) {
    const bool valid = kvrow < len;
#pragma unroll
    for (int i = 0; i < DBLK; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * i + 8 * g + 4 * hi;
        uint2 wk = make_uint2(0, 0), wv = make_uint2(0, 0);
        if (valid) {
          wk = make_uint2(pack2<DT>(dkacc[i][4 * g + 0], dkacc[i][4 * g + 1]), pack2<DT>(dkacc[i][4 * g + 2], dkacc[i][4 * g + 3]));
          wv = make_uint2(pack2<DT>(dvacc[i][4 * g + 0], dvacc[i][4 * g + 1]), pack2<DT>(dvacc[i][4 * g + 2], dvacc[i][4 * g + 3]));
        }
        *(uint2*)(dkp + d) = wk;
        *(uint2*)(dvp + d) = wv;
      }
  }
}

// ------------------------------------------------------------------------------------------------
// Kernel B: dQ
// ------------------------------------------------------------------------------------------------
template <int DT, int D, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_k(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CPR = D / 8;
  constexpr int TILE = 32 * D * 2;
  constexpr int STAGE = 3 * TILE;  // K, V, K^T
  constexpr int KSTEPS = D / 16, DBLK = D / 32;
  constexpr int NLD = TILE / (256 * 16);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int qblk = CAUSAL ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
  const int h = blockIdx.y, b = blockIdx.z;
  const int S = a.S;
  const int len = a.seqlens ? min(a.seqlens[b], S) : S;
  const int q0 = qblk * 128;
  const int qw0 = q0 + wave * 32;
  const int qrow = qw0 + l31;
  uint16_t* dqp = a.dq + ((int64_t)b * S + qrow) * a.lddq + (int64_t)h * D;

  if (q0 >= len) {
    if (qrow < S)
      for (int d = hi * (D / 2); d < (hi + 1) * (D / 2); d += 4) *(uint2*)(dqp + d) = make_uint2(0, 0);
    return;
  }

  uint4 qf[KSTEPS], dof[KSTEPS];
  {
    const int qr = min(qrow, S - 1);
    const uint16_t* qp = a.q + ((int64_t)b * S + qr) * a.ldq + (int64_t)h * D + 8 * hi;
    const uint16_t* dp = a.dout + ((int64_t)b * S + qr) * a.lddo + (int64_t)h * D + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      qf[ks] = *(const uint4*)(qp + 16 * ks);
      dof[ks] = *(const uint4*)(dp + 16 * ks);
    }
  }
  const int64_t bh = (int64_t)b * a.H + h;
  const int qsafe = min(qrow, S - 1);
  const float lse2 = a.lse[bh * a.S_pad + qsafe] * 1.4426950408889634f;
  const float dl = a.delta[bh * a.S_pad + qsafe];

  const int kv_end = CAUSAL ? min(len, q0 + 128) : len;
  const int ntiles = (kv_end + 31) / 32;

  int rrow[NLD], rcol[NLD], trow[NLD], tcol[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int qd = i * 256 + tid;
    rrow[i] = qd / CPR;
    rcol[i] = ((qd % CPR) ^ RSwz<D>::f(rrow[i])) * 8;
    trow[i] = qd >> 2;
    tcol[i] = ((qd & 3) ^ tswz(trow[i])) * 8;
  }
  auto stage = [&](int s, int kv0) {
    char* base = smem + s * STAGE;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int kr = min(kv0 + rrow[i], S - 1);
      const int64_t roff = ((int64_t)b * S + kr);
      glds16(a.k + roff * a.ldk + (int64_t)h * D + rcol[i], base + 0 * TILE + (i * 256 + wave * 64) * 16);
      glds16(a.v + roff * a.ldv + (int64_t)h * D + rcol[i], base + 1 * TILE + (i * 256 + wave * 64) * 16);
      glds16(a.kt + (bh * D + trow[i]) * a.S_pad + kv0 + tcol[i], base + 2 * TILE + (i * 256 + wave * 64) * 16);
    }
  };

  f32x16_t dqacc[DBLK];
#pragma unroll
  for (int i = 0; i < DBLK; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqacc[i][r] = 0.f;

  const int r_off = l31 * (D * 2);
  const int r_swz = RSwz<D>::f(l31);
  const float sc = a.scale_log2;

  stage(0, 0);
  for (int j = 0; j < ntiles; ++j) {
    const int kv0 = j * 32;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (j + 1 < ntiles) stage((j + 1) & 1, kv0 + 32);
    if (CAUSAL && kv0 > qw0 + 31) continue;
    const char* sK = smem + (j & 1) * STAGE;
    const char* sV = sK + TILE;
    const char* sKT = sK + 2 * TILE;

    f32x16_t sacc, pacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int coff = ((2 * ks + hi) ^ r_swz) << 4;
      const uint4 ka = *(const uint4*)(sK + r_off + coff);
      const uint4 va = *(const uint4*)(sV + r_off + coff);
      sacc = mfma32<DT>(ka, qf[ks], sacc);    // S^T[kv, q]
      pacc = mfma32<DT>(va, dof[ks], pacc);   // dP^T[kv, q]
    }
    float dsv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kv = kv0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const bool ok = (kv < len) && (qrow < len) && (!CAUSAL || kv <= qrow);
      const float p = ok ? exp2f(sacc[r] * sc - lse2) : 0.f;
      dsv[r] = ok ? p * (pacc[r] - dl) * a.scale : 0.f;
    }
    uint4 dsf[2];
    dsf[0] = pack8<DT>(dsv);
    dsf[1] = pack8<DT>(dsv + 8);
#pragma unroll
    for (int i = 0; i < DBLK; ++i) {
      const int row = 32 * i + l31;
      const int tsw = tswz(row);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const uint4 kt_a = *(const uint4*)(sKT + row * 64 + (((2 * s + hi) ^ tsw) << 4));
        dqacc[i] = mfma32<DT>(kt_a, dsf[s], dqacc[i]);
      }
    }
  }

  if (qrow < S) {
    const bool valid = qrow < len;
#pragma unroll
    for (int i = 0; i < DBLK; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * i + 8 * g + 4 * hi;
        uint2 w = make_uint2(0, 0);
        if (valid)
          w = make_uint2(pack2<DT>(dqacc[i][4 * g + 0], dqacc[i][4 * g + 1]), pack2<DT>(dqacc[i][4 * g + 2], dqacc[i][4 * g + 3]));
        *(uint2*)(dqp + d) = w;
      }
  }
}

template <int DT, int D, bool CAUSAL>
int launch_bwd(const BwdArgs& a, hipStream_t st) {
  constexpr size_t ldsA = 2 * 4 * (32 * D * 2), ldsB = 2 * 3 * (32 * D * 2);
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)attn_bwd_dkdv_k<DT, D, CAUSAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsA);
    hipFuncSetAttribute((const void*)attn_bwd_dq_k<DT, D, CAUSAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB);
    attr = true;
  }
  const int64_t nth = (int64_t)a.B * a.S * a.H;
  hipLaunchKernelGGL((delta_k<DT, D>), dim3((unsigned)((nth + 3) / 4)), dim3(256), 0, st, a);
  dim3 grid((a.S + 127) / 128, a.H, a.B);
  hipLaunchKernelGGL((attn_bwd_dkdv_k<DT, D, CAUSAL>), grid, dim3(256), ldsA, st, a);
  hipLaunchKernelGGL((attn_bwd_dq_k<DT, D, CAUSAL>), grid, dim3(256), ldsB, st, a);
  MH_LAUNCH_CHECK();
}

}  // namespace

extern "C" int64_t mh_attn_bwd_ws_elems(int B, int S, int H, int D) {
  const int64_t S_pad = (S + 63) / 64 * 64;
  return 3 * (int64_t)B * H * D * S_pad;  // Q^T, dO^T, K^T (16-bit elements)
}

extern "C" int mh_attn_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                           const void* o, int64_t ldo, const void* dout, int64_t lddo, const float* lse, float* delta,
                           void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, void* ws,
                           const int32_t* seqlens, int B, int S, int H, int D, int causal, int dt, void* stream) {
  if (!q || !k || !v || !o || !dout || !lse || !delta || !dq || !dk || !dv || !ws) return MH_ERR_ARG;
  if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (lddo & 7) || (ldo & 1) || (lddq & 3) || (lddk & 3) || (lddv & 3)) return MH_ERR_ARG;
  if (D != 128 && D != 64) return MH_ERR_SHAPE;
  const int S_pad = (S + 63) / 64 * 64;
  const int64_t one = (int64_t)B * H * D * S_pad;
  uint16_t* qt = (uint16_t*)ws;
  uint16_t* dot = qt + one;
  uint16_t* kt = dot + one;
  int rc;
  if ((rc = mh_attn_prep_v(q, ldq, qt, B, S, H, D, dt, stream)) != 0) return rc;
  if ((rc = mh_attn_prep_v(dout, lddo, dot, B, S, H, D, dt, stream)) != 0) return rc;
  if ((rc = mh_attn_prep_v(k, ldk, kt, B, S, H, D, dt, stream)) != 0) return rc;
  BwdArgs a;
  a.q = (const uint16_t*)q; a.k = (const uint16_t*)k; a.v = (const uint16_t*)v; a.o = (const uint16_t*)o;
  a.dout = (const uint16_t*)dout; a.qt = qt; a.dot = dot; a.kt = kt; a.lse = lse; a.delta = delta;
  a.dq = (uint16_t*)dq; a.dk = (uint16_t*)dk; a.dv = (uint16_t*)dv; a.seqlens = seqlens;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddo = lddo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  a.B = B; a.S = S; a.H = H; a.S_pad = S_pad;
  a.scale = 1.0f / sqrtf((float)D);
  a.scale_log2 = a.scale * 1.4426950408889634f;
  hipStream_t st = as_stream(stream);
#define GO(DT_, D_, C_) return launch_bwd<DT_, D_, C_>(a, st)
  if (dt == MH_BF16) {
    if (D == 128) { if (causal) GO(MH_BF16, 128, true); else GO(MH_BF16, 128, false); }
    else { if (causal) GO(MH_BF16, 64, true); else GO(MH_BF16, 64, false); }
  } else if (dt == MH_F16) {
    if (D == 128) { if (causal) GO(MH_F16, 128, true); else GO(MH_F16, 128, false); }
    else { if (causal) GO(MH_F16, 64, true); else GO(MH_F16, 64, false); }
  }
#undef GO
  return MH_ERR_DTYPE;
}
