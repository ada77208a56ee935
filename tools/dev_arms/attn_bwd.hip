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
#include "attn_bwd_common.h"

using namespace mhattn;

namespace {

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
  if (lane == 0) {
    const int64_t i = ((int64_t)b * a.H + h) * a.S_pad + s;
    a.delta[i] = acc;
    ((float*)a.lse2)[i] = a.lse[i] * 1.4426950408889634f;  // exp2-domain log-sum-exp for the kernels below
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
  const int nq = (a.S + 127) / 128;
  int bh_, qi;
  if (!xcd_work(a.B * a.H, nq, bh_, qi)) return;
  const int qblk = CAUSAL ? nq - 1 - qi : qi;
  const int h = bh_ % a.H, b = bh_ / a.H;
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
  const float lse2 = a.lse2[bh * a.S_pad + qsafe];
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
    for (int r = 0; r < 16; ++r) dsv[r] = fast_exp2(fmaf(sacc[r], sc, -lse2));
    if ((kv0 + 32 > len) || (qw0 + 32 > len) || (CAUSAL && (kv0 + 31 > qw0))) {  // wave-uniform: boundary tiles only
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kv = kv0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const bool ok = (kv < len) && (qrow < len) && (!CAUSAL || kv <= qrow);
        dsv[r] = ok ? dsv[r] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) dsv[r] *= (pacc[r] - dl);  // softmax scale is applied once, in the epilogue
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
          w = make_uint2(pack2<DT>(dqacc[i][4 * g + 0] * a.scale, dqacc[i][4 * g + 1] * a.scale), pack2<DT>(dqacc[i][4 * g + 2] * a.scale, dqacc[i][4 * g + 3] * a.scale));
        *(uint2*)(dqp + d) = w;
      }
  }
}

template <int DT, int D, bool CAUSAL>
int launch_bwd(const BwdArgs& a, int dt, hipStream_t st) {
  constexpr size_t ldsB = 2 * 3 * (32 * D * 2);
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)attn_bwd_dq_k<DT, D, CAUSAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB);
    attr = true;
  }
  const int64_t nth = (int64_t)a.B * a.S * a.H;
  hipLaunchKernelGGL((delta_k<DT, D>), dim3((unsigned)((nth + 3) / 4)), dim3(256), 0, st, a);
  const int rc = launch_attn_bwd_kv(a, dt, D, CAUSAL ? 1 : 0, st);
  if (rc != 0) return rc;
  dim3 grid(xcd_grid(a.B * a.H, (a.S + 127) / 128));
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
  a.dout = (const uint16_t*)dout; a.qt = qt; a.dot = dot; a.kt = kt; a.lse = lse; a.delta = delta; a.lse2 = delta + (int64_t)B * H * ((S + 63) / 64 * 64);
  a.dq = (uint16_t*)dq; a.dk = (uint16_t*)dk; a.dv = (uint16_t*)dv; a.seqlens = seqlens;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddo = lddo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  a.B = B; a.S = S; a.H = H; a.S_pad = S_pad;
  a.scale = 1.0f / sqrtf((float)D);
  a.scale_log2 = a.scale * 1.4426950408889634f;
  hipStream_t st = as_stream(stream);
#define GO(DT_, D_, C_) return launch_bwd<DT_, D_, C_>(a, dt, st)
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
