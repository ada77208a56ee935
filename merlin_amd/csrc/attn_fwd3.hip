// Flash attention forward, PING-PONG form (gfx950, D = 128): the two waves of a SIMD are kept in opposite phases.
//
// Same mathematics, tile machinery (attn_tiles.h), MFMA orientation and per-wave work as attn_fwd2.hip
//     S^T[kv, q] = K Q^T,   O^T[d, q] += V^T P^T   (32x32x16; 32 queries per wave, one query row per lane of a half-wave)
// but a block is 8 waves (256 queries) on one CU instead of two independent 4-wave blocks.  Measured on attn_fwd2
// (profiles/r03_attn_fwd_pingpong.txt): going from one to two free-running waves per SIMD buys only 19 % - the two waves fall into
// step, both in their MFMA section or both in their softmax section.  Here a wave's tile loop is cut into
//     M(t) = { O += V(t-1) P(t-1) ;  S(t) = K(t) Q^T }     32 MFMAs, no VALU work to speak of
//     V(t) = { softmax of S(t) -> P(t), rescale }          VALU / transcendental only, no MFMA, no LDS
// with a workgroup barrier behind every section, and waves 4-7 (the SIMD partners of waves 0-3) run one section behind: while one
// wave of a SIMD issues MFMAs the other issues the exponentials, by construction.
//
// K|V tiles go through a ring of three LDS stages (tile t+1 is requested one whole iteration before its first use: the stage it
// replaces was last read in M(t-1), and the lagging group leaves M(t-1) one barrier after the leading one); waves 0-3 copy the K
// half of a stage, waves 4-7 the V half.
//
// Result (same file): in shader cycles the overlap is there (3240 cycles per tile pair against 2750 for the M sections alone and
// 2650 for the V sections alone), but the part is power-limited and answers the denser instruction mix with a lower clock (1.65 GHz
// against 2.07 / 2.11 GHz for either half alone): +5 % without a mask, -3 % causal (256-query blocks leave a coarser diagonal).
// Outputs are bit-identical to attn_fwd2's.  Opt-in A/B arm: mh_attn_fwd_pingpong(1) selects it for D = 128; default off.
#include "attn_tiles.h"

#ifndef PP_PROBE
#define PP_PROBE 0  // development: 1 = no V sections, 2 = no M sections, 3 = no staging, 4 = cycles per tile into lse (timing only, wrong results)
#endif

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
__device__ __forceinline__ float max_halves(float x) {  // max over lane, lane ^ 32 without LDS traffic (the lgkm counter belongs to the fragment windows)
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float sum_halves(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
constexpr int PP_W = 4;  // fragment read-ahead of a section M
// LDS operations that may still be in flight when fragment g of a section is consumed: the next PP_W-1 fragments
// (fragments [0, NV) are transposed V fragments = two reads each, [NV, NF) K row fragments = one read each)
template <int NV, int NF>
constexpr int pp_allowed(int g) {
  int n = 0;
  for (int h = g + 1; h <= g + PP_W - 1 && h < NF; ++h) n += (h < NV) ? 2 : 1;
  return n;
}

template <int DT, bool CAUSAL>
__global__ __launch_bounds__(512, 1) void attn_fwd3_k(Fwd3Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int D = 128;
  constexpr int RB = D * 2;               // bytes per tile row
  constexpr int T_BYTES = 64 * RB;        // one [64][D] tile
  constexpr int STAGE = 2 * T_BYTES;      // K, V
  constexpr int KSTEPS = D / 16, DBLK = D / 32;
  constexpr int QROWS = 256;              // query rows per block

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;              // 0: leading group (waves 0-3), 1: their SIMD partners (wave w+4 shares a SIMD with wave w:
                                          // measured - any other split runs 20 % slower), one section behind
  const int role = wave >> 2;             // which half of a stage the wave copies (0: K, 1: V)
  const int l31 = lane & 31, hi = lane >> 5;
  const int nq = (a.S + QROWS - 1) / QROWS;
  int bh, qi;
  if (!xcd_work(a.B * a.H, nq, bh, qi)) return;
  const int qblk = CAUSAL ? nq - 1 - qi : qi;  // causal: heaviest q-blocks first
  const int h = bh % a.H, b = bh / a.H;
  const int S = a.S;
  const int len = a.seqlens ? min(a.seqlens[b], S) : S;
  const int q0 = qblk * QROWS;
  const int qw0 = q0 + wave * 32;
  const int qrow = qw0 + l31;

  if (q0 >= len) {  // whole block is padding: zeros (pad_input semantics)
    if (qrow < S) {
      uint16_t* op = a.o + ((int64_t)b * S + qrow) * a.ldo + (int64_t)h * D;
      for (int d = hi * (D / 2); d < (hi + 1) * (D / 2); d += 4) *(uint2*)(op + d) = make_uint2(0, 0);
      if (hi == 0) a.lse[((int64_t)b * a.H + h) * a.S_pad + qrow] = 0.f;
    }
    return;
  }
  const int kv_end = CAUSAL ? min(len, q0 + QROWS) : len;
  const int ntiles = (kv_end + 63) / 64;                           // K|V tiles the block stages (every wave takes part)

  // Q fragments (B operand of S^T): lane holds Q[qrow][16*ks + 8*hi .. +8]
  u32x4_t qf[KSTEPS];
  {
    const uint16_t* qp = a.q + ((int64_t)b * S + min(qrow, S - 1)) * a.ldq + (int64_t)h * D + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) qf[ks] = *(const u32x4_t*)(qp + 16 * ks);
  }
  // waves 0-3 copy the K tile of a stage, waves 4-7 its V tile
  const uint16_t* cbase = (role == 0 ? a.k + (int64_t)b * S * a.ldk : a.v + (int64_t)b * S * a.ldv) + (int64_t)h * D;
  const int64_t cld = role == 0 ? a.ldk : a.ldv;
  const auto so_c = stage_offsets<D, 64>(cld, tid & 255);
  auto stage = [&](int t) {
    stage_rows<D, 64>(cbase, cld, t * 64, S - 1, smem + (t % 3) * STAGE + role * T_BYTES, tid & 255, wave & 3, so_c);
  };

  f32x16_t o[DBLK];
#pragma unroll
  for (int i = 0; i < DBLK; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float sc = a.scale_log2;

  const unsigned lds0 = lds_addr_of(smem);
  unsigned off_k[KSTEPS], off_v[KSTEPS];  // KSTEPS == 2*DBLK
  row_frag_offsets<D>(l31, hi, off_k);
  tr_frag_offsets<D>(lane, off_v);

  f32x16_t st[2];      // S(t) of the wave's 32 queries x 64 keys, then P(t)
  u32x4_t pf[4];       // P(t) as four B-operand fragments
  u32x2_t wv[2 * PP_W];  // fragment read-ahead windows of a section M (a V^T fragment = two transpose-reads; assembled AFTER its wait)
  u32x4_t wk[PP_W];

  // fragment g of section M(t): g < NV -> V^T fragment (d-block g/4, k-step g%4) of tile t-1; else K row fragment n = g - NV
  // (key block n/8, k-step n%8) of tile t
  auto issue = [&](auto G, auto NV_, unsigned kb, unsigned vb) __attribute__((always_inline)) {
    constexpr int g = decltype(G)::value, NV = decltype(NV_)::value;
    if constexpr (g < NV) {
      lds_read64_tr<((g % 4) * 16) * RB>(wv[2 * (g % PP_W)], vb + off_v[2 * (g / 4)]);
      lds_read64_tr<((g % 4) * 16 + 8) * RB>(wv[2 * (g % PP_W) + 1], vb + off_v[2 * (g / 4) + 1]);
    } else {
      constexpr int n = g - NV;
      lds_read128<(n / KSTEPS) * 32 * RB>(wk[g % PP_W], kb + off_k[n % KSTEPS]);
    }
  };
  // M(t): O += V(t-1) P(t-1), then S(t) = K(t) Q^T.  One form only: M(0) multiplies the zero-filled stage 2 by P = 0, and the
  // S half of a wave's last M (t = nt_w) reads a stale stage and is never looked at.
  auto section_m = [&](int t) __attribute__((always_inline)) {
    constexpr int NV = 16, NF = 32;
    const unsigned kb = lds0 + (unsigned)(t % 3) * STAGE, vb = lds0 + (unsigned)((t + 2) % 3) * STAGE + T_BYTES;
    static_for<PP_W>([&](auto G) { issue(G, std::integral_constant<int, NV>{}, kb, vb); });
    prio_mfma(true);
    static_for<NF>([&](auto G) {
      constexpr int g = decltype(G)::value;
      lgkm_wait3<pp_allowed<NV, NF>(g)>();
      if constexpr (g < NV) {
        const u32x4_t vf = u32x4_t{wv[2 * (g % PP_W)][0], wv[2 * (g % PP_W)][1], wv[2 * (g % PP_W) + 1][0], wv[2 * (g % PP_W) + 1][1]};
        o[g / 4] = mfma32v<DT>(vf, pf[g % 4], o[g / 4]);
      } else {
        constexpr int n = g - NV;
        if constexpr (n % KSTEPS == 0) {  // (the score accumulators are born here, not at the top of the section: 32 registers)
#pragma unroll
          for (int r = 0; r < 16; ++r) st[n / KSTEPS][r] = 0.f;
        }
        st[n / KSTEPS] = mfma32v<DT>(wk[g % PP_W], qf[n % KSTEPS], st[n / KSTEPS]);
      }
      if constexpr (g + PP_W < NF) issue(std::integral_constant<int, g + PP_W>{}, std::integral_constant<int, NV>{}, kb, vb);
    });
    prio_mfma(false);
  };
  // mask (boundary tiles only) + online softmax of S(t): one query row per lane, lane ^ 32 holds the other 32 keys of the tile
  auto section_v = [&](int t) __attribute__((always_inline)) {
    const int kv0 = t * 64;
    if ((kv0 + 64 > len) || (CAUSAL && (kv0 + 63 > qw0))) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + 32 * blk + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const bool ok = (kv < len) && (!CAUSAL || kv <= qrow);
          st[blk][r] = ok ? st[blk][r] : -INFINITY;
        }
    }
    float mx = st[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[1][r]);
    mx = max_halves(mx);
    // lazy rescaling (see attn_fwd2.hip): the reference only moves when a row maximum grows by more than 2^8
    const float m_new = fmaxf(m_run, mx * sc);
    const bool need = m_new > m_run + 8.0f;
    float alpha = 1.0f;
    if (need) {
      alpha = fast_exp2(m_run - m_new);  // (first tile: exp2(-inf) = 0)
      m_run = m_new;
    }
    const float m_use = (m_run == -INFINITY) ? 0.f : m_run;  // rows with every key masked so far
    float psum = 0.f;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(fmaf(st[blk][r], sc, -m_use));
        st[blk][r] = p;
        psum += p;
      }
    l_run = l_run * alpha + psum;
    if (__builtin_amdgcn_ballot_w64(need) != 0) {  // wave-uniform, rare after the first tiles
#pragma unroll
      for (int i = 0; i < DBLK; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    }
    // P fragments: k-step s uses regs 8*(s&1)..+7 of key block s>>1
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float t8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t8[e] = st[s >> 1][8 * (s & 1) + e];
      pf[s] = pack8v<DT>(t8);
    }
  };
  auto bar = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto landed = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) pf[s4] = u32x4_t{0u, 0u, 0u, 0u};
  for (int i = tid; i < T_BYTES / 16; i += 512) *(uint4*)(smem + 2 * STAGE + T_BYTES + i * 16) = make_uint4(0, 0, 0, 0);  // "V(-1)"
  stage(0);
  if (ntiles > 1) stage(1);
  landed();
  bar();
  // Barrier n (n = 1, 2, ...): the leading group passes it after M(t) (n = 2t+1) and after V(t) (n = 2t+2), the lagging group before
  // M(0) (n = 1), after M(t) (n = 2t+2) and after V(t) (n = 2t+3).  The stage of tile t-2 was last read in M(t-1), which both groups
  // have left at barrier 2t: tile t+1 is requested behind it and has landed, for every wave, at barrier 2t+2.
  // Both groups run M(0) V(0) M(1) V(1) ... M(ntiles) with a barrier behind every section; the lagging group starts one barrier late.
  // Every wave computes on every tile of the block: the tiles above a wave's own causal diagonal are fully masked (P = 0) and cost
  // nothing - the wave would wait at the barriers anyway - and sections without conditions keep the register allocation simple.
  const long long pp_t0 = (PP_PROBE >= 4) ? __builtin_readcyclecounter() : 0;
  if (grp) bar();
  for (int t = 0; t < ntiles; ++t) {
    if (PP_PROBE != 3 && !grp && t >= 1 && t + 1 < ntiles) stage(t + 1);
    if (PP_PROBE != 2 && PP_PROBE != 6) section_m(t);
    if (grp) landed();
    bar();
    if (PP_PROBE != 3 && grp && t + 2 < ntiles) stage(t + 2);
    if (PP_PROBE != 1 && PP_PROBE != 5) section_v(t);
    if (!grp) landed();
    bar();
  }
  section_m(ntiles);  // the last tile's P V (its S half reads a stale stage and is never looked at)
  if (!grp) bar();
  const long long pp_t1 = (PP_PROBE >= 4) ? __builtin_readcyclecounter() : 0;

  // ---- finalize ----
  const float l_tot = sum_halves(l_run);
  const bool valid = (qrow < len);
  const float inv = (valid && l_tot > 0.f) ? 1.0f / l_tot : 0.f;
  if (qrow < S) {
    uint16_t* op = a.o + ((int64_t)b * S + qrow) * a.ldo + (int64_t)h * D;
#pragma unroll
    for (int i = 0; i < DBLK; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * i + 8 * g + 4 * hi;
        *(uint2*)(op + d) = make_uint2(pack2<DT>(o[i][4 * g + 0] * inv, o[i][4 * g + 1] * inv),
                                       pack2<DT>(o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv));
      }
    if (hi == 0)
      a.lse[((int64_t)b * a.H + h) * a.S_pad + qrow] = (valid && l_tot > 0.f) ? (m_run + log2f(l_tot)) * 0.6931471805599453f : 0.f;
  }
  if (PP_PROBE >= 4 && tid == 0) a.lse[((int64_t)b * a.H + h) * a.S_pad + q0] = (float)(pp_t1 - pp_t0) / (float)ntiles;
}

template <int DT, bool CAUSAL>
int launch_fwd3(const Fwd3Args& a, hipStream_t st) {
  constexpr size_t lds = 3 * 2 * 64 * 128 * 2;  // ring of three K|V stages
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)attn_fwd3_k<DT, CAUSAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL((attn_fwd3_k<DT, CAUSAL>), dim3(xcd_grid(a.B * a.H, (a.S + 255) / 256)), dim3(512), lds, st, a);
  MH_LAUNCH_CHECK();
}

}  // namespace

// D = 128 forward in the ping-pong form (called by mh_attn_fwd2 when mh_attn_fwd_pingpong(1) is set)
int launch_attn_fwd_pingpong(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo, float* lse,
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
