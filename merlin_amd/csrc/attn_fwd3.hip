// Flash attention forward, WIDE form (gfx950, D = 128): ONE wave per SIMD, 64 queries per wave.
//
// Same mathematics, tile machinery (attn_tiles.h) and MFMA orientation as attn_fwd2.hip
//     S^T[kv, q] = K Q^T,   O^T[d, q] += V^T P^T   (32x32x16; a lane owns one query row of each 32-query block)
// but a block is 4 waves x 64 queries = 256 query rows and runs alone on its CU with the whole 512-register file:
//   * every K row fragment and every transposed V fragment read from LDS feeds TWO MFMAs (the wave's two 32-query blocks): half the
//     LDS bytes per flop of the 32-query form, and a K|V tile is staged once per 256 instead of 128 queries;
//   * register classes are fixed by hand (inline-asm MFMAs): O accumulators and the Q fragments live in AccVGPRs, score accumulators
//     and P fragments in VGPRs, so the softmax arithmetic never moves data between the two files (left to hipcc, every accumulator of a
//     kernel that needs AccVGPRs at all sits there and each score is copied out first);
//   * the first MFMA of a score chain takes C = 0: no zero-fill of the 64 score registers per tile.
// mh_attn_fwd_wide(1) selects it for D = 128 (A/B against attn_fwd2: profiles/r03_attn_fwd_wide_ab.txt).
#include "attn_tiles.h"

namespace mhattn {
namespace {

struct Fwd3Args {
  const uint16_t *q, *k, *v;
  uint16_t* o;
  float* lse;
  const int32_t* seqlens;
  int64_t ldq, ldk, ldv, ldo;
  int B, S, H, S_pad;
  float scale_log2;
};

template <int N>
__device__ __forceinline__ void lgkm_wait3() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// score MFMAs: accumulator in VGPRs, B operand (Q fragment) in AccVGPRs; FIRST = start the chain from C = 0
template <int DT, bool FIRST>
__device__ __forceinline__ void mfma_s(f32x16_t& acc, const u32x4_t& kfrag, const u32x4_t& qfrag) {
  if constexpr (FIRST) {
    if constexpr (DT == MH_BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(kfrag), "a"(qfrag));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(kfrag), "a"(qfrag));
  } else {
    if constexpr (DT == MH_BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(kfrag), "a"(qfrag));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(kfrag), "a"(qfrag));
  }
}
// output MFMAs: accumulator pinned in AccVGPRs
template <int DT>
__device__ __forceinline__ void mfma_o(f32x16_t& acc, const u32x4_t& vfrag, const u32x4_t& pfrag) {
  if constexpr (DT == MH_BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(vfrag), "v"(pfrag));
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(vfrag), "v"(pfrag));
}
// hipcc cannot see that the asm statements are MFMAs: the wait states it would insert are written out
__device__ __forceinline__ void settle_v(f32x16_t& a, f32x16_t& b, f32x16_t& c, f32x16_t& d) {  // last score MFMA -> first VALU read
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void ready_p(u32x4_t (&p)[2][4]) {  // VALU write -> MFMA read (every P fragment is an operand: none may be packed later)
  asm volatile("s_nop 1" : "+v"(p[0][0]), "+v"(p[0][1]), "+v"(p[0][2]), "+v"(p[0][3]), "+v"(p[1][0]), "+v"(p[1][1]), "+v"(p[1][2]), "+v"(p[1][3]));
}
__device__ __forceinline__ void settle_a(f32x16_t& a) { asm volatile("s_nop 15\n\ts_nop 3" : "+a"(a)); }
__device__ __forceinline__ void drain_o(f32x16_t& a, f32x16_t& b, f32x16_t& c, f32x16_t& d, f32x16_t& e, f32x16_t& f, f32x16_t& g, f32x16_t& h) {
  asm volatile("s_nop 15\n\ts_nop 3" : "+a"(a), "+a"(b), "+a"(c), "+a"(d), "+a"(e), "+a"(f), "+a"(g), "+a"(h));
}
template <int OFF>
__device__ __forceinline__ void gload128_a(u32x4_t& d, const void* ptr) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(d) : "v"(ptr), "n"(OFF) : "memory");
}

template <int DT, bool CAUSAL>
__global__ __launch_bounds__(256, 1) void attn_fwd3_k(Fwd3Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int D = 128;
  constexpr int RB = D * 2;               // bytes per tile row
  constexpr int T_BYTES = 64 * RB;        // one [64][D] tile
  constexpr int STAGE = 2 * T_BYTES;      // K, V
  constexpr int KSTEPS = D / 16, DBLK = D / 32;
  constexpr int NKF = 2 * KSTEPS;         // K fragments per tile (2 key blocks x 8 k-steps)
  constexpr int WK = 8;                   // K read window (fragments)
  constexpr int NVF = DBLK * 4;           // V^T fragments per tile (d-block i, k-step s)
  constexpr int QROWS = 256;              // query rows per block

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int nq = (a.S + QROWS - 1) / QROWS;
  int bh, qi;
  if (!xcd_work(a.B * a.H, nq, bh, qi)) return;
  const int qblk = CAUSAL ? nq - 1 - qi : qi;  // causal: heaviest q-blocks first
  const int h = bh % a.H, b = bh / a.H;
  const int S = a.S;
  const int len = a.seqlens ? min(a.seqlens[b], S) : S;
  const int q0 = qblk * QROWS;
  const int qw0 = q0 + wave * 64;
  const int qrow0 = qw0 + l31, qrow1 = qrow0 + 32;

  if (q0 >= len) {  // whole block is padding: zeros (pad_input semantics)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qr = qrow0 + 32 * qb;
      if (qr < S) {
        uint16_t* op = a.o + ((int64_t)b * S + qr) * a.ldo + (int64_t)h * D;
        for (int d = hi * (D / 2); d < (hi + 1) * (D / 2); d += 4) *(uint2*)(op + d) = make_uint2(0, 0);
        if (hi == 0) a.lse[((int64_t)b * a.H + h) * a.S_pad + qr] = 0.f;
      }
    }
    return;
  }
  const int kv_end = CAUSAL ? min(len, q0 + QROWS) : len;
  const int ntiles = (kv_end + 63) / 64;

  // Q fragments (B operand of S^T) straight into AccVGPRs: lane holds Q[qrow][16*ks + 8*hi .. +8]; waited for by the first tile's vmcnt(0)
  u32x4_t qf[2][KSTEPS];
  {
    const uint16_t* qp0 = a.q + ((int64_t)b * S + min(qrow0, S - 1)) * a.ldq + (int64_t)h * D + 8 * hi;
    const uint16_t* qp1 = a.q + ((int64_t)b * S + min(qrow1, S - 1)) * a.ldq + (int64_t)h * D + 8 * hi;
    static_for<KSTEPS>([&](auto I) {
      constexpr int ks = decltype(I)::value;
      gload128_a<32 * ks>(qf[0][ks], qp0);
      gload128_a<32 * ks>(qf[1][ks], qp1);
    });
  }
  const uint16_t* kbase = a.k + (int64_t)b * S * a.ldk + (int64_t)h * D;
  const uint16_t* vbase = a.v + (int64_t)b * S * a.ldv + (int64_t)h * D;
  const auto so_k = stage_offsets<D, 64>(a.ldk, tid), so_v = stage_offsets<D, 64>(a.ldv, tid);
  auto stage = [&](int s, int kv0) {
    char* base = smem + s * STAGE;
    stage_rows<D, 64>(kbase, a.ldk, kv0, S - 1, base, tid, wave, so_k);
    stage_rows<D, 64>(vbase, a.ldv, kv0, S - 1, base + T_BYTES, tid, wave, so_v);
  };

  f32x16_t o[2][DBLK];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int i = 0; i < DBLK; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][i][r] = 0.f;
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  const float sc = a.scale_log2;

  const unsigned lds0 = lds_addr_of(smem);
  unsigned off_k[KSTEPS], off_v[KSTEPS];  // KSTEPS == 2*DBLK
  row_frag_offsets<D>(l31, hi, off_k);
  tr_frag_offsets<D>(lane, off_v);

  auto tile = [&](int j, auto EDGE_) {
    constexpr bool EDGE = decltype(EDGE_)::value;
    const int kv0 = j * 64;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (j + 1 < ntiles) stage((j + 1) & 1, kv0 + 64);
    if constexpr (EDGE) {
      if (CAUSAL && kv0 > qw0 + 63) return;  // tile entirely above this wave's diagonal (wave-uniform)
    }
    const unsigned sb = lds0 + (unsigned)(j & 1) * STAGE;
    unsigned ak[KSTEPS], av[KSTEPS];
#pragma unroll
    for (int i = 0; i < KSTEPS; ++i) {
      ak[i] = sb + off_k[i];
      av[i] = sb + T_BYTES + off_v[i];
    }

    // ---- S^T = K Q^T: fragment n = (key block n / KSTEPS, k-step n % KSTEPS) feeds the two query blocks; rolling window of WK reads ----
    f32x16_t st[2][2];  // [query block][key block]
    u32x4_t wk[WK];
    static_for<WK>([&](auto I) {
      constexpr int n = decltype(I)::value;
      lds_read128<(n / KSTEPS) * 32 * RB>(wk[n % WK], ak[n % KSTEPS]);
    });
    static_for<NKF>([&](auto I) {
      constexpr int n = decltype(I)::value;
      constexpr int left = NKF - 1 - n;
      lgkm_wait3<(left < WK - 1 ? left : WK - 1)>();
      mfma_s<DT, (n % KSTEPS) == 0>(st[0][n / KSTEPS], wk[n % WK], qf[0][n % KSTEPS]);
      mfma_s<DT, (n % KSTEPS) == 0>(st[1][n / KSTEPS], wk[n % WK], qf[1][n % KSTEPS]);
      if constexpr (n + WK < NKF) lds_read128<((n + WK) / KSTEPS) * 32 * RB>(wk[n % WK], ak[(n + WK) % KSTEPS]);
    });

    // ---- first V^T fragments go out now; their latency hides under the softmax ----
    u32x2_t wv[8];  // window of 4 fragments = 8 transpose-reads
    static_for<4>([&](auto I) {
      constexpr int f = decltype(I)::value;  // f = i*4 + s
      lds_read64_tr<((f % 4) * 16) * RB>(wv[2 * f], av[2 * (f / 4)]);
      lds_read64_tr<((f % 4) * 16 + 8) * RB>(wv[2 * f + 1], av[2 * (f / 4) + 1]);
    });
    settle_v(st[0][0], st[0][1], st[1][0], st[1][1]);

    // ---- mask (boundary tiles only) + online softmax per query block (one query row per lane; lane^32 holds the other 32 keys) ----
    u32x4_t pf[2][4];
    float alpha[2];
    bool need_any = false;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qrow = qrow0 + 32 * qb;
      if (EDGE && ((kv0 + 64 > len) || (CAUSAL && (kv0 + 63 > qw0 + 32 * qb)))) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kv = kv0 + 32 * blk + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const bool ok = (kv < len) && (!CAUSAL || kv <= qrow);
            st[qb][blk][r] = ok ? st[qb][blk][r] : -INFINITY;
          }
      }
      float mx = st[qb][0][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[qb][0][r]);
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[qb][1][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      // lazy rescaling (see attn_fwd2.hip): the reference only moves when a row maximum grows by more than 2^8
      const float m_new = fmaxf(m_run[qb], mx * sc);
      const bool need = m_new > m_run[qb] + 8.0f;
      alpha[qb] = 1.0f;
      if (need) {
        alpha[qb] = fast_exp2(m_run[qb] - m_new);  // (first tile: exp2(-inf) = 0)
        m_run[qb] = m_new;
      }
      need_any |= need;
      const float m_use = (m_run[qb] == -INFINITY) ? 0.f : m_run[qb];  // rows with every key masked so far
      float psum = 0.f;
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = fast_exp2(fmaf(st[qb][blk][r], sc, -m_use));
          st[qb][blk][r] = p;
          psum += p;
        }
      l_run[qb] = l_run[qb] * alpha[qb] + psum;
      // P fragments: k-step s uses regs 8*(s&1)..+7 of key block s>>1
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = st[qb][s >> 1][8 * (s & 1) + e];
        pf[qb][s] = pack8v<DT>(t);
      }
    }
    if (__builtin_amdgcn_ballot_w64(need_any) != 0) {  // wave-uniform, rare after the first tiles
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int i = 0; i < DBLK; ++i) {
          settle_a(o[qb][i]);
#pragma unroll
          for (int r = 0; r < 16; ++r) o[qb][i][r] *= alpha[qb];
          asm volatile("s_nop 1" : "+a"(o[qb][i]));
        }
    }
    ready_p(pf);
    __builtin_amdgcn_sched_barrier(0);

    // ---- O^T += V^T P^T: fragment f = (d-block f/4, k-step f%4) feeds the two query blocks; rolling window of 4 fragments ----
    static_for<NVF>([&](auto I) {
      constexpr int f = decltype(I)::value;
      constexpr int left = NVF - 1 - f;
      lgkm_wait3<2 * (left < 3 ? left : 3)>();
      const u32x4_t vf = u32x4_t{wv[2 * (f % 4)][0], wv[2 * (f % 4)][1], wv[2 * (f % 4) + 1][0], wv[2 * (f % 4) + 1][1]};
      mfma_o<DT>(o[0][f / 4], vf, pf[0][f % 4]);
      mfma_o<DT>(o[1][f / 4], vf, pf[1][f % 4]);
      if constexpr (f + 4 < NVF) {
        constexpr int g = f + 4;
        lds_read64_tr<((g % 4) * 16) * RB>(wv[2 * (f % 4)], av[2 * (g / 4)]);
        lds_read64_tr<((g % 4) * 16 + 8) * RB>(wv[2 * (f % 4) + 1], av[2 * (g / 4) + 1]);
      }
    });
    // the register allocator moves accumulators between the two loops and the finalize (AccVGPR copies on the loop-exit
    // edge, straight behind the last MFMA whose latency it cannot see): every tile ends with the MFMA -> VALU wait states
    static_assert(DBLK == 4, "drain_o lists the accumulators");
    drain_o(o[0][0], o[0][1], o[0][2], o[0][3], o[1][0], o[1][1], o[1][2], o[1][3]);
  };
  // tiles [0, n_full) need no masking for any wave of this block
  const int n_full = min(ntiles, CAUSAL ? min(q0, len) / 64 : len / 64);
  stage(0, 0);
  for (int j = 0; j < n_full; ++j) tile(j, std::false_type{});
  for (int j = n_full; j < ntiles; ++j) tile(j, std::true_type{});

  // ---- finalize ----
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = qrow0 + 32 * qb;
    const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
    const bool valid = (qrow < len);
    const float inv = (valid && l_tot > 0.f) ? 1.0f / l_tot : 0.f;
#pragma unroll
    for (int i = 0; i < DBLK; ++i) settle_a(o[qb][i]);
    if (qrow < S) {
      uint16_t* op = a.o + ((int64_t)b * S + qrow) * a.ldo + (int64_t)h * D;
#pragma unroll
      for (int i = 0; i < DBLK; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = 32 * i + 8 * g + 4 * hi;
          *(uint2*)(op + d) = make_uint2(pack2<DT>(o[qb][i][4 * g + 0] * inv, o[qb][i][4 * g + 1] * inv),
                                         pack2<DT>(o[qb][i][4 * g + 2] * inv, o[qb][i][4 * g + 3] * inv));
        }
      if (hi == 0)
        a.lse[((int64_t)b * a.H + h) * a.S_pad + qrow] = (valid && l_tot > 0.f) ? (m_run[qb] + log2f(l_tot)) * 0.6931471805599453f : 0.f;
    }
  }
}

template <int DT, bool CAUSAL>
int launch_fwd3(const Fwd3Args& a, hipStream_t st) {
  constexpr size_t lds = 2 * 2 * 64 * 128 * 2;
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)attn_fwd3_k<DT, CAUSAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL((attn_fwd3_k<DT, CAUSAL>), dim3(xcd_grid(a.B * a.H, (a.S + 255) / 256)), dim3(256), lds, st, a);
  MH_LAUNCH_CHECK();
}

}  // namespace

// D = 128 forward in the wide form (called by mh_attn_fwd2 when mh_attn_fwd_wide(1) is set)
int launch_attn_fwd_wide(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo, float* lse,
                         const int32_t* seqlens, int B, int S, int H, int causal, int dt, hipStream_t st) {
  Fwd3Args a;
  a.q = (const uint16_t*)q; a.k = (const uint16_t*)k; a.v = (const uint16_t*)v; a.o = (uint16_t*)o;
  a.lse = lse; a.seqlens = seqlens; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.B = B; a.S = S; a.H = H; a.S_pad = (S + 63) / 64 * 64;
  a.scale_log2 = (1.0f / sqrtf(128.0f)) * 1.4426950408889634f;
  if (dt == MH_BF16) return causal ? launch_fwd3<MH_BF16, true>(a, st) : launch_fwd3<MH_BF16, false>(a, st);
  if (dt == MH_F16) return causal ? launch_fwd3<MH_F16, true>(a, st) : launch_fwd3<MH_F16, false>(a, st);
  return MH_ERR_DTYPE;
}

}  // namespace mhattn
