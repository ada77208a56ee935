// Flash attention forward for gfx950 (causal D=128 for Llama, non-causal D=64 for CLIP).
//
// Replaces flash_attn_varlen_qkvpacked_func(causal=True) + unpad_input/pad_input
// (mmgpt/utils/llama_flash_attn_monkey_patch.py:68-102) and CLIP's eager softmax attention.
//
// Design (wave64 / MFMA 32x32x16 first, not a warp-32 tiling):
//   * block = 4 waves, each wave owns 32 query rows (block = 128 rows); K/V tiles of 64 keys are
//     staged once per block in LDS with global_load_lds (double buffered, one barrier per tile).
//   * TRANSPOSED products so that every softmax quantity is lane-local:
//       S^T[kv, q] = K * Q^T   : A operand = K rows from LDS (ds_read_b128), B operand = Q, held in
//                                registers for the whole kernel (lane: q = lane&31, 8 d's / k-step).
//                                Each lane then holds 32 scores of ONE query row (the other 32 are in
//                                lane^32): row max = 31 in-lane fmax + one cross-half exchange.
//       O^T[d, q]  = V^T * P^T : A operand = V^T rows from LDS, B operand = P straight from the S^T
//                                accumulator registers (8 consecutive regs -> one bf16x8 fragment): NO
//                                cross-lane movement of P at all.  The MFMA contraction index is a free
//                                permutation; the one implied by the S^T register layout is baked into
//                                V^T's key order by mh_attn_prep_v (bits 2 and 3 of the key index
//                                swapped inside every group of 16 keys), so V^T fragments are plain
//                                16-byte LDS reads.
//   * online softmax in the exp2 domain, fp32 statistics; per-lane alpha (one query row per lane).
//   * LDS bank swizzle on the global source address + the ds_read address (rule 21):
//       256-B rows (K, D=128): chunk ^= row & 15;   128-B rows (K D=64, V^T): chunk ^= (row>>1) & 7.
//   * varlen: keys >= seqlens[b] are masked, query rows >= seqlens[b] are written as zeros
//     (pad_input semantics).  With right padding + causal masking valid rows never see padded keys.
//   * causal: q-blocks are launched heaviest-first; a wave skips tiles entirely above its diagonal.
#include "mh_common.h"
#include "merlin_hip_dev.h"

namespace {

struct AttnArgs {
  const uint16_t* q;
  const uint16_t* k;
  const uint16_t* vt;
  uint16_t* o;
  float* lse;
  const int32_t* seqlens;
  int64_t ldq, ldk, ldo;
  int B, S, H, S_pad;
  float scale_log2;  // softmax scale * log2(e)
};

template <int D> struct KSwz;
template <> struct KSwz<128> { static __device__ __forceinline__ int f(int row) { return row & 15; } };
template <> struct KSwz<64> { static __device__ __forceinline__ int f(int row) { return (row >> 1) & 7; } };

template <int DT, int D, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_fwd_k(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KCPR = D / 8;             // 16-B chunks per K row
  constexpr int K_BYTES = 64 * D * 2;     // K tile [64][D]
  constexpr int V_BYTES = D * 64 * 2;     // V^T tile [D][64]
  constexpr int STAGE = K_BYTES + V_BYTES;
  constexpr int KSTEPS = D / 16;          // QK^T k-steps
  constexpr int DBLK = D / 32;            // O^T row blocks
  constexpr int NLD = K_BYTES / (256 * 16);  // glds per thread per tile (K and V each)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nq = (a.S + 127) / 128;
  int bh, qi;
  if (!xcd_work(a.B * a.H, nq, bh, qi)) return;
  const int qblk = CAUSAL ? nq - 1 - qi : qi;  // causal: heaviest q-blocks first
  const int h = bh % a.H, b = bh / a.H;
  const int S = a.S;
  const int len = a.seqlens ? min(a.seqlens[b], S) : S;
  const int q0 = qblk * 128;
  const int qw0 = q0 + wave * 32;   // first query row of this wave
  const int qrow = qw0 + l31;       // this lane's query row

  if (q0 >= len) {  // whole block is padding: zeros
    if (qrow < S) {
      uint16_t* op = a.o + ((int64_t)b * S + qrow) * a.ldo + (int64_t)h * D;
      for (int d = hi * (D / 2); d < (hi + 1) * (D / 2); d += 8) *(uint4*)(op + d) = make_uint4(0, 0, 0, 0);
      if (hi == 0) a.lse[((int64_t)b * a.H + h) * a.S_pad + qrow] = 0.f;
    }
    return;
  }
  const int kv_end = CAUSAL ? min(len, q0 + 128) : len;
  const int ntiles = (kv_end + 63) / 64;

  // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[qrow][16*st + 8*hi .. +8] ----
  uint4 qf[KSTEPS];
  {
    const int qr = min(qrow, S - 1);
    const uint16_t* qp = a.q + ((int64_t)b * S + qr) * a.ldq + (int64_t)h * D + 8 * hi;
#pragma unroll
    for (int st = 0; st < KSTEPS; ++st) qf[st] = *(const uint4*)(qp + 16 * st);
  }

  // ---- staging sources ----
  const uint16_t* ksrc[NLD];
  const uint16_t* vsrc[NLD];
  int krow_l[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int qd = i * 256 + tid;
    {
      const int row = qd / KCPR, cc = qd % KCPR;
      const int c = cc ^ KSwz<D>::f(row);
      krow_l[i] = row;
      ksrc[i] = a.k + (int64_t)h * D + c * 8;  // + (b*S + kv0 + row) * ldk per tile (row clamped)
    }
    {
      const int row = qd >> 3, cc = qd & 7;  // row = d
      const int c = cc ^ ((row >> 1) & 7);
      vsrc[i] = a.vt + (((int64_t)b * a.H + h) * D + row) * a.S_pad + c * 8;  // + kv0 per tile
    }
  }
  auto stage = [&](int s, int kv0) {
    char* sK = smem + s * STAGE;
    char* sV = sK + K_BYTES;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int kr = min(kv0 + krow_l[i], S - 1);
      glds16(ksrc[i] + ((int64_t)b * S + kr) * a.ldk, sK + (i * 256 + wave * 64) * 16);
      glds16(vsrc[i] + kv0, sV + (i * 256 + wave * 64) * 16);
    }
  };

  f32x16_t o[DBLK];
#pragma unroll
  for (int i = 0; i < DBLK; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float sc = a.scale_log2;

  // per-lane LDS read offsets
  const int k_row_off = l31 * (D * 2);
  const int k_swz = KSwz<D>::f(l31);
  const int v_swz = (l31 >> 1) & 7;  // row = 32*dblk + l31 -> (row>>1)&7 == (l31>>1)&7 since 32*dblk is a multiple of 16

  stage(0, 0);
  for (int j = 0; j < ntiles; ++j) {
    const int kv0 = j * 64;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (j + 1 < ntiles) stage((j + 1) & 1, kv0 + 64);
    if (CAUSAL && kv0 > qw0 + 31) continue;  // tile entirely above this wave's diagonal (wave-uniform)
    const char* sK = smem + (j & 1) * STAGE;
    const char* sV = sK + K_BYTES;

    // ---- S^T = K Q^T : two 32-key blocks ----
    f32x16_t st[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[blk][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const uint4 kf = *(const uint4*)(sK + blk * 32 * (D * 2) + k_row_off + (((2 * ks + hi) ^ k_swz) << 4));
        st[blk] = mfma32<DT>(kf, qf[ks], st[blk]);
      }
    }
    // ---- mask (only on tiles that need it) ----
    const bool need_mask = (kv0 + 64 > len) || (CAUSAL && (kv0 + 63 > qw0));
    if (need_mask) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + 32 * blk + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const bool ok = (kv < len) && (!CAUSAL || kv <= qrow);
          if (!ok) st[blk][r] = -INFINITY;
        }
    }
    // ---- online softmax (one query row per lane; partner lane^32 holds the other 32 keys) ----
    float mx = st[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[1][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * sc);
    // rows whose every key so far is masked keep m = -inf; guard the subtraction
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = fast_exp2(m_run - m_use);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(st[blk][r] * sc - m_use);
        st[blk][r] = p;
        psum += p;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int i = 0; i < DBLK; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    // ---- P fragments: step s uses regs 8*(s&1)..+7 of block s>>1 ----
    uint4 pf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = st[s >> 1][8 * (s & 1) + e];
      pf[s] = pack8<DT>(t);
    }
    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int i = 0; i < DBLK; ++i) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const uint4 vf = *(const uint4*)(sV + (32 * i + l31) * 128 + (((2 * s + hi) ^ v_swz) << 4));
        o[i] = mfma32<DT>(vf, pf[s], o[i]);
      }
    }
  }

  // ---- finalize ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const bool valid = (qrow < len);
  const float inv = (valid && l_tot > 0.f) ? 1.0f / l_tot : 0.f;
  if (qrow < S) {
    uint16_t* op = a.o + ((int64_t)b * S + qrow) * a.ldo + (int64_t)h * D;
#pragma unroll
    for (int i = 0; i < DBLK; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * i + 8 * g + 4 * hi;
        const uint2 w = make_uint2(pack2<DT>(o[i][4 * g + 0] * inv, o[i][4 * g + 1] * inv),
                                   pack2<DT>(o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv));
        *(uint2*)(op + d) = w;
      }
    if (hi == 0)
      a.lse[((int64_t)b * a.H + h) * a.S_pad + qrow] = (valid && l_tot > 0.f) ? (m_run + log2f(l_tot)) * 0.6931471805599453f : 0.f;
  }
}

// vt[b, h, d, perm(s)] = v[(b*S + s)*ldv + h*D + d];  perm swaps bits 2 and 3 of s.  Tail [S, S_pad) zero.
template <int D>
__global__ __launch_bounds__(256) void prep_v_k(const uint16_t* __restrict__ v, int64_t ldv, uint16_t* __restrict__ vt,
                                                int B, int S, int H, int S_pad) {
  __shared__ uint16_t t[64][D + 2];
  const int s0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  // load 64 keys x D (coalesced along d)
  for (int i = threadIdx.x; i < 64 * D; i += 256) {
    const int s = i / D, d = i % D;
    t[s][d] = (s0 + s < S) ? v[((int64_t)b * S + s0 + s) * ldv + (int64_t)h * D + d] : (uint16_t)0;
  }
  __syncthreads();
  // write D rows x 64 positions (coalesced along position)
  for (int i = threadIdx.x; i < 64 * D; i += 256) {
    const int d = i / 64, p = i % 64;
    // position p holds key s = p with bits 2,3 swapped (the swap is an involution)
    const int s = (p & ~12) | ((p & 4) << 1) | ((p & 8) >> 1);
    vt[(((int64_t)b * H + h) * D + d) * S_pad + s0 + p] = t[s][d];
  }
}

template <int DT, int D, bool CAUSAL>
int launch_fwd(const AttnArgs& a, hipStream_t st) {
  constexpr size_t lds = 2 * (64 * D * 2 + D * 64 * 2);
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)attn_fwd_k<DT, D, CAUSAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  dim3 grid(xcd_grid(a.B * a.H, (a.S + 127) / 128));
  hipLaunchKernelGGL((attn_fwd_k<DT, D, CAUSAL>), grid, dim3(256), lds, st, a);
  MH_LAUNCH_CHECK();
}

}  // namespace

extern "C" int mh_attn_prep_v(const void* v, int64_t ldv, void* vt, int B, int S, int H, int D, int dt, void* stream) {
  if (!v || !vt || B <= 0 || S <= 0 || H <= 0) return MH_ERR_ARG;
  (void)dt;
  const int S_pad = (S + 63) / 64 * 64;
  dim3 grid(S_pad / 64, H, B);
  if (D == 128)
    hipLaunchKernelGGL(prep_v_k<128>, grid, dim3(256), 0, as_stream(stream), (const uint16_t*)v, ldv, (uint16_t*)vt, B, S, H, S_pad);
  else if (D == 64)
    hipLaunchKernelGGL(prep_v_k<64>, grid, dim3(256), 0, as_stream(stream), (const uint16_t*)v, ldv, (uint16_t*)vt, B, S, H, S_pad);
  else return MH_ERR_SHAPE;
  MH_LAUNCH_CHECK();
}

extern "C" int mh_attn_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, void* o, int64_t ldo,
                           float* lse, const int32_t* seqlens, int B, int S, int H, int D, int causal, int dt, void* stream) {
  if (!q || !k || !vt || !o || !lse || B <= 0 || S <= 0 || H <= 0) return MH_ERR_ARG;
  if ((ldq & 7) || (ldk & 7) || (ldo & 7) || !aligned16(o) || !aligned16(q) || !aligned16(k) || !aligned16(vt)) return MH_ERR_ARG;
  AttnArgs a;
  a.q = (const uint16_t*)q; a.k = (const uint16_t*)k; a.vt = (const uint16_t*)vt; a.o = (uint16_t*)o;
  a.lse = lse; a.seqlens = seqlens; a.ldq = ldq; a.ldk = ldk; a.ldo = ldo;
  a.B = B; a.S = S; a.H = H; a.S_pad = (S + 63) / 64 * 64;
  a.scale_log2 = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  hipStream_t st = as_stream(stream);
  if (dt == MH_BF16) {
    if (D == 128 && causal) return launch_fwd<MH_BF16, 128, true>(a, st);
    if (D == 128 && !causal) return launch_fwd<MH_BF16, 128, false>(a, st);
    if (D == 64 && causal) return launch_fwd<MH_BF16, 64, true>(a, st);
    if (D == 64 && !causal) return launch_fwd<MH_BF16, 64, false>(a, st);
    return MH_ERR_SHAPE;
  } else if (dt == MH_F16) {
    if (D == 128 && causal) return launch_fwd<MH_F16, 128, true>(a, st);
    if (D == 128 && !causal) return launch_fwd<MH_F16, 128, false>(a, st);
    if (D == 64 && causal) return launch_fwd<MH_F16, 64, true>(a, st);
    if (D == 64 && !causal) return launch_fwd<MH_F16, 64, false>(a, st);
    return MH_ERR_SHAPE;
  }
  return MH_ERR_DTYPE;
}
