// Flash attention forward v2 (gfx950): same math and MFMA orientation as attn_fwd.hip
//     S^T[kv, q] = K Q^T  (row fragments of the K tile x Q fragments held in registers)
//     O^T[d, q]  = V^T P^T (TRANSPOSED fragments of the row-major V tile x P straight from the S^T accumulators)
// but with
//   * V consumed ROW-MAJOR through ds_read_b64_tr_b16 transpose-reads (attn_tiles.h): no mh_attn_prep_v pass, no V^T copy;
//   * all LDS fragment reads as hand-waited inline asm in a rolling 8-deep window (counted lgkmcnt), so (a) the next
//     tile's global_load_lds really stay in flight under the whole tile (hipcc would drain them before the first
//     ds_read it can see) and (b) LDS latency is covered by the MFMAs of the same wave;
//   * the first P*V fragment reads are issued BEFORE the softmax VALU block.
// Replaces flash_attn_varlen_qkvpacked_func(causal=True) + unpad_input/pad_input
// (mmgpt/utils/llama_flash_attn_monkey_patch.py:68-102) and CLIP's eager softmax attention.
#include "attn_tiles.h"

namespace mhattn {
namespace {

struct Fwd2Args {
  const uint16_t *q, *k, *v;
  uint16_t* o;
  float* lse;
  const int32_t* seqlens;
  int64_t ldq, ldk, ldv, ldo;
  int B, S, H, S_pad;
  float scale_log2;
  int wide;  // output rows are 16-byte aligned: the epilogue stores 16 bytes per lane (attn_tiles.h, store_row_wide)
};

template <int N>
__device__ __forceinline__ void lgkm_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <int DT, int D, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_fwd2_k(Fwd2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RB = D * 2;               // bytes per tile row
  constexpr int T_BYTES = 64 * RB;        // one [64][D] tile
  constexpr int STAGE = 2 * T_BYTES;      // K, V
  constexpr int KSTEPS = D / 16, DBLK = D / 32;
  constexpr int NKF = 2 * KSTEPS;         // K fragments per tile (2 key blocks)
  constexpr int WK = NKF < 8 ? NKF : 8;   // read window (fragments)
  constexpr int NVF = DBLK * 4;           // V^T fragments per tile (d-block i, k-step s)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int nq = (a.S + 127) / 128;
  int bh, qi;
  if (!xcd_work(a.B * a.H, nq, bh, qi)) return;
  const int qblk = CAUSAL ? nq - 1 - qi : qi;  // causal: heaviest q-blocks first
  const int h = bh % a.H, b = bh / a.H;
  const int S = a.S;
  const int len = a.seqlens ? min(a.seqlens[b], S) : S;
  const int q0 = qblk * 128;
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
  const int kv_end = CAUSAL ? min(len, q0 + 128) : len;
  const int ntiles = (kv_end + 63) / 64;

  // Q fragments (B operand of S^T): lane holds Q[qrow][16*ks + 8*hi .. +8]
  u32x4_t qf[KSTEPS];
  {
    const uint16_t* qp = a.q + ((int64_t)b * S + min(qrow, S - 1)) * a.ldq + (int64_t)h * D + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) qf[ks] = *(const u32x4_t*)(qp + 16 * ks);
  }
  const uint16_t* kbase = a.k + (int64_t)b * S * a.ldk + (int64_t)h * D;
  const uint16_t* vbase = a.v + (int64_t)b * S * a.ldv + (int64_t)h * D;
  const unsigned lds0 = lds_addr_of(smem);
  const auto src_k = row_src<D>(kbase, a.ldk, S, tid), src_v = row_src<D>(vbase, a.ldv, S, tid);
  auto stage = [&](int s, int kv0) {  // scalar addressing only (attn_tiles.h, stage_rows_buf); rows >= S arrive as zeros and are masked below
    const unsigned base = lds0 + (unsigned)s * STAGE + (unsigned)wave * 1024u;
    stage_rows_buf<D, 64>(src_k, kv0, base);
    stage_rows_buf<D, 64>(src_v, kv0, base + T_BYTES);
  };

  f32x16_t o[DBLK];
#pragma unroll
  for (int i = 0; i < DBLK; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float sc = a.scale_log2;

  unsigned off_k[KSTEPS], off_v[KSTEPS];  // KSTEPS == 2*DBLK
  row_frag_offsets<D>(l31, hi, off_k);
  tr_frag_offsets<D>(lane, off_v);

  // One KV tile.  EDGE = false: the tile is fully visible to every wave of the block (no mask, no skip) - the
  // common case runs branch-free; EDGE = true: tiles on the causal diagonal / at the sequence end.
  // fragment addresses of stage 0; the stage a tile reads (PAR = j & 1, a template parameter: the tile loop is unrolled by two) goes into the
  // ds_read immediates, so no address is computed per tile
  unsigned ak[KSTEPS], av[KSTEPS];
#pragma unroll
  for (int i = 0; i < KSTEPS; ++i) {
    ak[i] = lds0 + off_k[i];
    av[i] = lds0 + T_BYTES + off_v[i];
  }
  static_assert(STAGE + T_BYTES + 56 * RB < 65536, "stage offset + fragment offset must fit the 16-bit ds_read immediate");
  auto tile = [&](int j, auto EDGE_, auto PAR_) {
    constexpr bool EDGE = decltype(EDGE_)::value;
    constexpr int SO = decltype(PAR_)::value * STAGE;  // byte offset of the stage this tile reads
    const int kv0 = j * 64;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (j + 1 < ntiles) stage(1 - decltype(PAR_)::value, kv0 + 64);
    if constexpr (EDGE) {
      if (CAUSAL && kv0 > qw0 + 31) return;  // tile entirely above this wave's diagonal (wave-uniform)
    }

    // ---- S^T = K Q^T: fragment n = (key block n / KSTEPS, k-step n % KSTEPS), rolling window of WK reads ----
    f32x16_t st[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[0][r] = 0.f; st[1][r] = 0.f; }
    u32x4_t wk[WK];
    static_for<WK>([&](auto I) {
      constexpr int n = decltype(I)::value;
      lds_read128<SO + (n / KSTEPS) * 32 * RB>(wk[n % WK], ak[n % KSTEPS]);
    });
    prio_mfma(true);
    static_for<NKF>([&](auto I) {
      constexpr int n = decltype(I)::value;
      constexpr int left = NKF - 1 - n;
      lgkm_wait<(left < WK - 1 ? left : WK - 1)>();
      st[n / KSTEPS] = mfma32v<DT>(wk[n % WK], qf[n % KSTEPS], st[n / KSTEPS]);
      if constexpr (n + WK < NKF) lds_read128<SO + ((n + WK) / KSTEPS) * 32 * RB>(wk[n % WK], ak[(n + WK) % KSTEPS]);
    });

    prio_mfma(false);
    // ---- first V^T fragments go out now; their latency hides under the softmax ----
    u32x2_t wv[8];  // window of 4 fragments = 8 transpose-reads
    static_for<4>([&](auto I) {
      constexpr int f = decltype(I)::value;  // f = i*4 + s
      lds_read64_tr<SO + ((f % 4) * 16) * RB>(wv[2 * f], av[2 * (f / 4)]);
      lds_read64_tr<SO + ((f % 4) * 16 + 8) * RB>(wv[2 * f + 1], av[2 * (f / 4) + 1]);
    });

    // ---- mask (boundary tiles only), online softmax (one query row per lane; lane^32 holds the other 32 keys) ----
    if (EDGE && ((kv0 + 64 > len) || (CAUSAL && (kv0 + 63 > qw0)))) {
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
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // Lazy rescaling: the running reference m_run only moves when the row maximum grows by more than 2^8 (P stays <= 256,
    // exact in fp32 accumulation and well inside the 16-bit range); the result is the same softmax - any reference works
    // as long as O and l use the same one - but after the first few tiles no lane needs a rescale and the 16*DBLK
    // multiplies of the O accumulators are skipped for the whole wave.
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
    if (__builtin_amdgcn_ballot_w64(need) != 0) {  // wave-uniform
#pragma unroll
      for (int i = 0; i < DBLK; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    }
    // P fragments: k-step s uses regs 8*(s&1)..+7 of key block s>>1
    u32x4_t pf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = st[s >> 1][8 * (s & 1) + e];
      pf[s] = pack8v<DT>(t);
    }

    // ---- O^T += V^T P^T: fragment f = (d-block f/4, k-step f%4), rolling window of 4 fragments ----
    prio_mfma(true);
    static_for<NVF>([&](auto I) {
      constexpr int f = decltype(I)::value;
      constexpr int left = NVF - 1 - f;
      lgkm_wait<2 * (left < 3 ? left : 3)>();
      const u32x4_t vf = u32x4_t{wv[2 * (f % 4)][0], wv[2 * (f % 4)][1], wv[2 * (f % 4) + 1][0], wv[2 * (f % 4) + 1][1]};
      o[f / 4] = mfma32v<DT>(vf, pf[f % 4], o[f / 4]);
      if constexpr (f + 4 < NVF) {
        constexpr int g = f + 4;
        lds_read64_tr<SO + ((g % 4) * 16) * RB>(wv[2 * (f % 4)], av[2 * (g / 4)]);
        lds_read64_tr<SO + ((g % 4) * 16 + 8) * RB>(wv[2 * (f % 4) + 1], av[2 * (g / 4) + 1]);
      }
    });
    prio_mfma(false);
  };
  // tiles [0, n_full) need no masking for any wave of this block
  const int n_full = min(ntiles, CAUSAL ? min(q0, len) / 64 : len / 64);
  stage(0, 0);
  // The Q fragments (plain global loads, waited for by hipcc) are USED here, once: left to their first use - inside the tile loop - hipcc
  // keeps a counted `s_waitcnt vmcnt(7) .. vmcnt(0)` in front of the first eight MFMAs of EVERY tile, and that counter also holds the
  // eight LDS-DMA copies of the next tile requested just above them: each wave sat out the whole copy it had just requested before it
  // finished its first score chain (the prefetch overlapped nothing within the wave).  Here the wait covers Q and tile 0 together.
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) asm volatile("" : "+v"(qf[ks]));
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  int j = 0;
  for (; j + 1 < n_full; j += 2) {
    tile(j, std::false_type{}, P0{});
    tile(j + 1, std::false_type{}, P1{});
  }
  if (j < n_full) tile(j++, std::false_type{}, P0{});  // (j is even here)
  for (; j < ntiles; ++j) {
    if (j & 1) tile(j, std::true_type{}, P1{});
    else tile(j, std::true_type{}, P0{});
  }

  // ---- finalize ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const bool valid = (qrow < len);
  const float inv = (valid && l_tot > 0.f) ? 1.0f / l_tot : 0.f;
  if (qrow < S) {
    uint16_t* op = a.o + ((int64_t)b * S + qrow) * a.ldo + (int64_t)h * D;
    auto val = [&](int i, int r) { return o[i][r] * inv; };
    if (a.wide) store_row_wide<DT, DBLK>(op, hi, true, val);  // 16-byte stores (T21); rows past `len` are written as zeros (inv = 0)
    else store_row_narrow<DT, DBLK>(op, hi, true, val);
    if (hi == 0)
      a.lse[((int64_t)b * a.H + h) * a.S_pad + qrow] = (valid && l_tot > 0.f) ? (m_run + log2f(l_tot)) * 0.6931471805599453f : 0.f;
  }
}

template <int DT, int D, bool CAUSAL>
int launch_fwd2(const Fwd2Args& a, hipStream_t st) {
  constexpr size_t lds = 2 * 2 * 64 * D * 2;
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)attn_fwd2_k<DT, D, CAUSAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL((attn_fwd2_k<DT, D, CAUSAL>), dim3(xcd_grid(a.B * a.H, (a.S + 127) / 128)), dim3(256), lds, st, a);
  MH_LAUNCH_CHECK();
}

}  // namespace
}  // namespace mhattn

namespace mhattn {
int launch_attn_fwd_pingpong(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo, float* lse,
                         const int32_t* seqlens, int B, int S, int H, int causal, int dt, hipStream_t st);  // attn_fwd3.hip
int launch_attn_fwd_wave64(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo, float* lse,
                           const int32_t* seqlens, int B, int S, int H, int causal, int dt, hipStream_t st);  // attn_fwd4.hip
int g_attn_wide_stores = 1;   // A/B switch (mh_attn_wide_stores): 16-byte epilogue stores in the forward / backward kernels
int g_attn_fwd_pingpong = 0;  // D = 128 forward form: 0 = attn_fwd2 (default), 1 = attn_fwd3 (ping-pong), 2 = attn_fwd4 (one wave per SIMD, 64 rows per wave)
}  // namespace mhattn
extern "C" void mh_attn_wide_stores(int on) { mhattn::g_attn_wide_stores = on ? 1 : 0; }
extern "C" void mh_attn_fwd_pingpong(int on) { mhattn::g_attn_fwd_pingpong = (on == 2) ? 2 : (on ? 1 : 0); }

extern "C" int mh_attn_fwd2(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o,
                            int64_t ldo, float* lse, const int32_t* seqlens, int B, int S, int H, int D, int causal, int dt,
                            void* stream) {
  using namespace mhattn;
  if (!q || !k || !v || !o || !lse || B <= 0 || S <= 0 || H <= 0) return MH_ERR_ARG;
  if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3) || !aligned16(q) || !aligned16(k) || !aligned16(v)) return MH_ERR_ARG;
  if ((int64_t)S * ldk * 2 >= (1ll << 31) || (int64_t)S * ldv * 2 >= (1ll << 31)) return MH_ERR_SHAPE;  // one batch element under a 31-bit num_records (stage_rows_buf)
  if (g_attn_fwd_pingpong == 2 && D == 128 && (dt == MH_BF16 || dt == MH_F16))
    return launch_attn_fwd_wave64(q, ldq, k, ldk, v, ldv, o, ldo, lse, seqlens, B, S, H, causal, dt, as_stream(stream));
  if (g_attn_fwd_pingpong && D == 128 && (dt == MH_BF16 || dt == MH_F16))
    return launch_attn_fwd_pingpong(q, ldq, k, ldk, v, ldv, o, ldo, lse, seqlens, B, S, H, causal, dt, as_stream(stream));
  Fwd2Args a;
  a.q = (const uint16_t*)q; a.k = (const uint16_t*)k; a.v = (const uint16_t*)v; a.o = (uint16_t*)o;
  a.lse = lse; a.seqlens = seqlens; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.B = B; a.S = S; a.H = H; a.S_pad = (S + 63) / 64 * 64;
  a.scale_log2 = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  a.wide = ((ldo & 7) == 0) && aligned16(o) && g_attn_wide_stores;
  hipStream_t st = as_stream(stream);
#define GO(DT_, D_, C_) return launch_fwd2<DT_, D_, C_>(a, st)
  if (dt == MH_BF16) {
    if (D == 128) { if (causal) GO(MH_BF16, 128, true); else GO(MH_BF16, 128, false); }
    if (D == 64) { if (causal) GO(MH_BF16, 64, true); else GO(MH_BF16, 64, false); }
    return MH_ERR_SHAPE;
  } else if (dt == MH_F16) {
    if (D == 128) { if (causal) GO(MH_F16, 128, true); else GO(MH_F16, 128, false); }
    if (D == 64) { if (causal) GO(MH_F16, 64, true); else GO(MH_F16, 64, false); }
    return MH_ERR_SHAPE;
  }
#undef GO
  return MH_ERR_DTYPE;
}
