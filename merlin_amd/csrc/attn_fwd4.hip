// Flash attention forward, ONE WAVE PER SIMD form (gfx950, D = 128): 4 waves x 64 query rows, the whole 512-register file per wave.
//
// Same mathematics, tile machinery (attn_tiles.h) and MFMA orientation as attn_fwd2.hip
//     S^T[kv, q] = K Q^T,   O^T[d, q] += V^T P^T   (32x32x16; one query row per lane of a half-wave)
// and the same per-row arithmetic in the same order (results are bit-identical to attn_fwd2's), but
//   * a wave owns TWO 32-row query halves: every K fragment (and, per product segment, every V^T fragment) that comes out of LDS
//     feeds the MFMAs of both halves' accumulators - K is read ONCE per tile into 64 AccVGPRs (ds_read_b128 straight into the
//     accumulator file) and used as the A operand from there; attn_fwd2 reads 1 KB of LDS per MFMA, this form 0.6 KB;
//   * the softmax arithmetic of one half is placed BETWEEN the MFMAs of the other half, by construction: with one wave per SIMD
//     nothing else can fill the matrix pipe's issue gaps (attn_fwd2 relies on the arbitration between two free-running waves of a
//     SIMD, which measured 52 % MFMA-busy).  A tile is four segments of 16 MFMAs:
//         A(t): S(h0,t) = K(t) Q0^T      ||  softmax(h1, t-1), second part (exp, row sums, packing)
//         B(t): O(h1) += V(t-1) P(h1,t-1) ||  softmax(h0, t), first part (mask, row maximum, running reference)
//         C(t): S(h1,t) = K(t) Q1^T      ||  softmax(h0, t), second part
//         D(t): O(h0) += V(t) P(h0,t)     ||  softmax(h1, t), first part   || K(t+1) -> AccVGPRs
//     every MFMA step carries a fixed slice of the other half's VALU work behind it (sched_barrier between steps);
//   * accumulator classes are fixed by inline asm (O: AccVGPRs, S: VGPRs, K fragments: AccVGPRs), as in the fused dK|dV kernel -
//     hipcc does not know these statements are MFMAs, so the wait states are part of the schedule: an S accumulator is first
//     read by VALU work at least two MFMAs of another chain (>= 64 cycles) after its last MFMA, explicit s_nops where a segment has none;
//   * K tiles are requested two tiles ahead and V tiles one ahead (a K tile's LDS slot is free as soon as it sits in registers), two
//     slots each = 64 KB of LDS, one barrier per tile.
// Opt-in A/B arm: mh_attn_fwd_pingpong(2) selects it for D = 128 (0: attn_fwd2, 1: attn_fwd3 ping-pong).
#include "attn_tiles.h"

// Two wait states in front of the MFMAs of every tile body but the steady one (SAFE): hipcc places AccVGPR copies (v_accvgpr_write /
// _mov: live-range splits, control-flow merges between the tile bodies) directly in front of these statements without knowing that
// they read the register as an MFMA does.  For the steady body tools/check_mfma_hazards.py on the -S listing must report no hazard.
namespace mhattn {
namespace {

struct Fwd4Args {
  const uint16_t *q, *k, *v;
  uint16_t* o;
  float* lse;
  const int32_t* seqlens;
  int64_t ldq, ldk, ldv, ldo;
  int B, S, H, S_pad;
  float scale_log2;
};

template <int N>
__device__ __forceinline__ void lgkm_wait4() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// S^T accumulators live in VGPRs, the K fragment (A operand) in AccVGPRs; the first k-step of a chain takes the inline-constant zero C.
// The second half's Q fragments (B operand) sit in AccVGPRs as well (QA): 192 of the 256 are otherwise idle and the 256 VGPRs are not
// enough for both halves' Q, two S^T tiles, the packed P and the fragment windows.  O^T accumulators are pinned in AccVGPRs.
#define F4_ASM(SAFE_, TXT, OUTS, INS)                          \
  do {                                                         \
    if constexpr (SAFE_) asm volatile("s_nop 1\n\t" TXT : OUTS : INS); \
    else asm volatile(TXT : OUTS : INS);                       \
  } while (0)
#define F4_C ,
template <int DT, bool SAFE, bool QA, bool ZERO>
__device__ __forceinline__ void mfma_sx(f32x16_t& acc, const u32x4_t& ka, const u32x4_t& q) {
  if constexpr (DT == MH_BF16) {
    if constexpr (ZERO && !QA) F4_ASM(SAFE, "v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0", "=&v"(acc), "a"(ka) F4_C "v"(q));
    if constexpr (ZERO && QA) F4_ASM(SAFE, "v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0", "=&v"(acc), "a"(ka) F4_C "a"(q));
    if constexpr (!ZERO && !QA) F4_ASM(SAFE, "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0", "+v"(acc), "a"(ka) F4_C "v"(q));
    if constexpr (!ZERO && QA) F4_ASM(SAFE, "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0", "+v"(acc), "a"(ka) F4_C "a"(q));
  } else {
    if constexpr (ZERO && !QA) F4_ASM(SAFE, "v_mfma_f32_32x32x16_f16 %0, %1, %2, 0", "=&v"(acc), "a"(ka) F4_C "v"(q));
    if constexpr (ZERO && QA) F4_ASM(SAFE, "v_mfma_f32_32x32x16_f16 %0, %1, %2, 0", "=&v"(acc), "a"(ka) F4_C "a"(q));
    if constexpr (!ZERO && !QA) F4_ASM(SAFE, "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0", "+v"(acc), "a"(ka) F4_C "v"(q));
    if constexpr (!ZERO && QA) F4_ASM(SAFE, "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0", "+v"(acc), "a"(ka) F4_C "a"(q));
  }
}
template <int DT, bool SAFE>
__device__ __forceinline__ void mfma_o(f32x16_t& acc, const u32x4_t& vf, const u32x4_t& p) {
  if constexpr (DT == MH_BF16) F4_ASM(SAFE, "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0", "+a"(acc), "v"(vf) F4_C "v"(p));
  else F4_ASM(SAFE, "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0", "+a"(acc), "v"(vf) F4_C "v"(p));
}
template <int OFF>
__device__ __forceinline__ void lds_read128_acc(u32x4_t& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(d) : "v"(addr), "n"(OFF));
}
// Pins of the hand-placed schedule.  sched_barrier only binds the machine scheduler: the IR passes and the DAG linearisation move pure
// arithmetic freely across asm statements and SINK it into later blocks (the first build had every exponential behind the segment it was
// written into).  A slice of VALU work is therefore bracketed by empty volatile asm statements: pin_in() redefines an input (nothing that
// depends on it can be computed earlier), pin_out() uses a result (it has to exist by then); volatile asm statements keep their order.
__device__ __forceinline__ void pin_in(float& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin_in_s(float& x) { asm volatile("" : "+s"(x)); }  // wave-uniform value: no VALU hazard is assumed behind a scalar definition
__device__ __forceinline__ void pin_in(f32x16_t& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin_out(const float& x) { asm volatile("" ::"v"(x)); }
__device__ __forceinline__ void pin_out(const u32x4_t& x) { asm volatile("" ::"v"(x)); }
__device__ __forceinline__ void pin_out(const f32x16_t& x) { asm volatile("" ::"v"(x)); }
// single-instruction maxima on values that come out of inline-asm MFMAs (fmaxf would first canonicalise every input: hipcc cannot
// see that an asm result is not a signalling NaN)
__device__ __forceinline__ float vmax2(float a, float b) {
  float d;
  asm volatile("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float d;
  asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float max_halves4(float x) {  // max over lane, lane ^ 32 without LDS traffic (the lgkm counter belongs to the fragment windows)
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return vmax2(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ void settle_s(f32x16_t& a, f32x16_t& b) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void settle_o(f32x16_t& a) { asm volatile("s_nop 15\n\ts_nop 3" : "+a"(a)); }

__device__ __forceinline__ void use_acc16(const u32x4_t (&x)[16]) {
  asm volatile("" ::"a"(x[0]), "a"(x[1]), "a"(x[2]), "a"(x[3]), "a"(x[4]), "a"(x[5]), "a"(x[6]), "a"(x[7]));
  asm volatile("" ::"a"(x[8]), "a"(x[9]), "a"(x[10]), "a"(x[11]), "a"(x[12]), "a"(x[13]), "a"(x[14]), "a"(x[15]));
}

__device__ __forceinline__ void settle_all(f32x16_t (&o)[2][4]) {
  asm volatile("" : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[0][2]), "+a"(o[0][3]), "+a"(o[1][0]), "+a"(o[1][1]), "+a"(o[1][2]), "+a"(o[1][3]));
}

constexpr int F4_W = 4;  // V^T fragments in flight in a product segment (two transpose-reads each)
// LDS operations issued after fragment f's reads and before the wait in front of MFMA step f of a product segment
// (program order: window fragments 0..W-1; step g: wait, MFMA g, fragment g + W, KR K-fragment reads)
constexpr int f4_allowed(int f, int KR) {
  int n = 0;
  if (f < F4_W) {
    n += 2 * (F4_W - 1 - f);
    for (int g = 0; g < f; ++g) n += (g + F4_W < 16 ? 2 : 0) + KR;
  } else {
    n += KR;  // the K reads of step f - W follow fragment f in that step
    for (int g = f - F4_W + 1; g < f; ++g) n += (g + F4_W < 16 ? 2 : 0) + KR;
  }
  return n;
}
constexpr int f4_e0(int k) { return k <= 10 ? 0 : ((k - 10) * 32) / 22; }  // softmax elements [e0(k), e0(k + 1)) are exponentiated in slice k

template <int DT, bool CAUSAL>
__global__ __launch_bounds__(256, 1) void attn_fwd4_k(Fwd4Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int D = 128;
  constexpr int RB = D * 2;          // bytes per tile row
  constexpr int T_BYTES = 64 * RB;   // one [64][D] tile; LDS: K slots 0 / 1, then V slots 0 / 1
  constexpr int KSTEPS = D / 16, DBLK = D / 32;
  constexpr int QROWS = 256;
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;

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
  const int qrow0 = qw0 + l31, qrow1 = qw0 + 32 + l31;

  if (q0 >= len) {  // whole block is padding: zeros (pad_input semantics)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int qrow = hh ? qrow1 : qrow0;
      if (qrow < S) {
        uint16_t* op = a.o + ((int64_t)b * S + qrow) * a.ldo + (int64_t)h * D;
        for (int d = hi * (D / 2); d < (hi + 1) * (D / 2); d += 4) *(uint2*)(op + d) = make_uint2(0, 0);
        if (hi == 0) a.lse[((int64_t)b * a.H + h) * a.S_pad + qrow] = 0.f;
      }
    }
    return;
  }
  const int kv_end = CAUSAL ? min(len, q0 + QROWS) : len;
  const int ntiles = (kv_end + 63) / 64;

  // Q fragments (B operand of S^T): lane holds Q[qrow][16*ks + 8*hi .. +8], both halves
  u32x4_t qf[2][KSTEPS];
  {
    const uint16_t* qp0 = a.q + ((int64_t)b * S + min(qrow0, S - 1)) * a.ldq + (int64_t)h * D + 8 * hi;
    const uint16_t* qp1 = a.q + ((int64_t)b * S + min(qrow1, S - 1)) * a.ldq + (int64_t)h * D + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      qf[0][ks] = *(const u32x4_t*)(qp0 + 16 * ks);
      qf[1][ks] = *(const u32x4_t*)(qp1 + 16 * ks);
    }
  }
  // the second half's Q fragments move into AccVGPRs HERE, once: as VGPR values hipcc copies them in front of every use, and a
  // v_accvgpr_write directly in front of an inline-asm MFMA that reads the register is a hazard it does not see
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) asm volatile("s_nop 1" : "+a"(qf[1][ks]));
  // ... and the first half's are USED here, before any tile copy is requested: hipcc otherwise waits for these loads at their first use
  // INSIDE the tile loop with a counted vmcnt that knows nothing of the LDS-DMA copies in flight (s_waitcnt vmcnt(3) in segment A = a
  // stall on the copies just requested, every tile)
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) asm volatile("" : "+v"(qf[0][ks]));
  const uint16_t* kbase = a.k + (int64_t)b * S * a.ldk + (int64_t)h * D;
  const uint16_t* vbase = a.v + (int64_t)b * S * a.ldv + (int64_t)h * D;
  const unsigned lds0 = lds_addr_of(smem);
  const auto src_k = row_src<D>(kbase, a.ldk, S, tid), src_v = row_src<D>(vbase, a.ldv, S, tid);
  // Tile copies (LDS-DMA, attn_tiles.h layout) without a branch and without per-tile vector work: ONE descriptor per tile - base at the
  // tile's first row, num_records = what is left of the batch element from there - and the row group of copy instruction i in a VGPR offset
  // (voff + i * 16 rows: the hardware range-checks base + voffset, so rows past the batch element arrive as zeros in every instruction,
  // also in a partial last tile; stage_rows_buf's scalar offsets take no part in that check and need a second path with a branch).
  // A tile past the end of the batch element gets num_records = 0: the copy is requested anyway (zeros into a free slot) so that the
  // request sits in straight-line code between the MFMAs of segment C.
  unsigned vo_k[4], vo_v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    vo_k[i] = src_k.voff + (i ? src_k.step[i - 1] : 0u);
    vo_v[i] = src_v.voff + (i ? src_v.step[i - 1] : 0u);
  }
  auto stage4 = [&](const RowSrc<D>& src, const unsigned (&vo)[4], int t, unsigned lds_wave) {
    const long long adv = (long long)t * 64 * (long long)src.row_bytes;
    const long long left = (long long)src.span - adv;
    const i32x4_t rs = row_srd(src.base + (uint64_t)adv, left > 0 ? (unsigned)left : 0u);
    unsigned keep;
    asm volatile(
        "s_nop 4\n\ts_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %1, 0 offen lds\n\t"
        "s_add_u32 m0, %2, 0x1000\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %1, 0 offen lds\n\t"
        "s_add_u32 m0, %2, 0x2000\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %1, 0 offen lds\n\t"
        "s_add_u32 m0, %2, 0x3000\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %1, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(rs), "s"(lds_wave), "v"(vo[0]), "v"(vo[1]), "v"(vo[2]), "v"(vo[3])
        : "memory", "scc");
  };
  auto stage_k = [&](int t) { stage4(src_k, vo_k, t, lds0 + (unsigned)(t & 1) * T_BYTES + (unsigned)wave * 1024u); };
  auto stage_v = [&](int t) { stage4(src_v, vo_v, t, lds0 + (unsigned)(2 + (t & 1)) * T_BYTES + (unsigned)wave * 1024u); };

  f32x16_t o[2][DBLK];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int i = 0; i < DBLK; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[hh][i][r] = 0.f;
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  float mxa[2], mxb[2], alpha[2] = {1.f, 1.f}, negm[2] = {0.f, 0.f}, psum[2] = {0.f, 0.f};
  bool need[2] = {false, false};
  const float sc = a.scale_log2;
  float pend[2] = {0.f, 0.f};  // the row sum runs one element behind its exponential (no wait state between v_exp_f32 and its consumer)

  unsigned off_k[KSTEPS], off_v[KSTEPS];
  row_frag_offsets<D>(l31, hi, off_k);
  tr_frag_offsets<D>(lane, off_v);
  unsigned ak[KSTEPS], av[KSTEPS];  // fragment addresses INCLUDING the slot (see `toggle` below)
#pragma unroll
  for (int i = 0; i < KSTEPS; ++i) {
    ak[i] = lds0 + off_k[i];
    av[i] = lds0 + 2 * T_BYTES + off_v[i];
  }
  static_assert(T_BYTES + 56 * RB + 32 * RB < 65536, "slot + fragment offset must fit the 16-bit ds_read immediate");

  f32x16_t st[2][2];   // S^T / P of the two halves: [half][key block]
  u32x4_t pf[2][4];    // packed P fragments: [half][k-step]
  u32x4_t ka[2 * KSTEPS];  // the K tile's 16 row fragments (key block n / KSTEPS, k-step n % KSTEPS), AccVGPRs

  // ---- one slice of a half's softmax (slices 0..15: the segment after its S^T is complete; 16..31: the segment after that) ----
  auto sm = [&](auto H_, auto EDGE_, auto K_, int kv0) {
    constexpr int H = decltype(H_)::value, k = decltype(K_)::value;
    constexpr bool EDGE = decltype(EDGE_)::value;
    const int qrow = H ? qrow1 : qrow0;
    if constexpr (k == 2 && EDGE) {  // mask, boundary tiles only - and there unconditionally: a (wave-uniform) branch inside a product segment
                                     // lets hipcc copy / spill around it registers whose LDS reads are still in flight
      {
        // key kv = kv0 + 32 blk + c_r + 4 hi (c_r = (r & 3) + 8 (r >> 2)) is visible iff kv < len and (causal) kv <= qrow, i.e. iff
        // c_r <= lim - 32 blk with ONE per-lane bound: a compare against an immediate and a select per element, two live registers
        const int vis = CAUSAL ? min(len - 1, qrow) : len - 1;
        const int lim0 = vis - kv0 - 4 * hi, lim1 = lim0 - 32;
        pin_in(st[H][0]);
        pin_in(st[H][1]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          st[H][0][r] = (c <= lim0) ? st[H][0][r] : -INFINITY;
          st[H][1][r] = (c <= lim1) ? st[H][1][r] : -INFINITY;
        }
        pin_in(st[H][0]);
        pin_in(st[H][1]);
      }
    }
    if constexpr (k >= 4 && k < 8) {  // row maximum: four instructions per slice, the two key blocks' chains alternate (a dependent pair of
                                      // asm statements back to back costs a wait state: hipcc assumes the worst of an asm definition)
      constexpr int j = k - 4;        // elements 4j .. 4j+3 of both key blocks
      const f32x16_t& s0 = st[H][0];
      const f32x16_t& s1 = st[H][1];
      float m0 = mxa[H], m1 = mxb[H];
      if constexpr (j == 0) {
        m0 = vmax2(s0[0], s0[1]);
        m1 = vmax2(s1[0], s1[1]);
        m0 = vmax3(m0, s0[2], s0[3]);
        m1 = vmax3(m1, s1[2], s1[3]);
      } else {
        m0 = vmax3(m0, s0[4 * j], s0[4 * j + 1]);
        m1 = vmax3(m1, s1[4 * j], s1[4 * j + 1]);
        m0 = vmax3(m0, s0[4 * j + 2], s0[4 * j + 3]);
        m1 = vmax3(m1, s1[4 * j + 2], s1[4 * j + 3]);
      }
      mxa[H] = m0;
      mxb[H] = m1;
    }
    if constexpr (k == 8) mxa[H] = max_halves4(vmax2(mxa[H], mxb[H]));
    if constexpr (k == 9) {
      // lazy rescaling exactly as in attn_fwd2: the running reference moves only when the row maximum grows by more than 2^8
      const float m_new = fmaxf(m_run[H], mxa[H] * sc);
      need[H] = m_new > m_run[H] + 8.0f;
      alpha[H] = 1.0f;
      if (need[H]) {
        alpha[H] = fast_exp2(m_run[H] - m_new);
        m_run[H] = m_new;
      }
      negm[H] = (m_run[H] == -INFINITY) ? 0.f : -m_run[H];
      psum[H] = 0.f;
      pin_out(alpha[H]);
      pin_out(negm[H]);
    }
    if constexpr (k >= 10) {
      constexpr int e_lo = f4_e0(k), e_hi = f4_e0(k + 1);
      pin_in(negm[H]);
      static_for<e_hi - e_lo>([&](auto E_) {
        constexpr int e = e_lo + decltype(E_)::value;
        const float p = fast_exp2(fmaf(st[H][e >> 4][e & 15], sc, negm[H]));
        st[H][e >> 4][e & 15] = p;
        if constexpr (e > 0) psum[H] += pend[H];
        pend[H] = p;
      });
      static_for<4>([&](auto S_) {  // P fragment s = registers 8*(s&1)..+7 of key block s>>1, packed as soon as its last element exists
        constexpr int s = decltype(S_)::value;
        if constexpr (8 * s + 7 >= e_lo && 8 * s + 7 < e_hi) {
          float t[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) t[e] = st[H][s >> 1][8 * (s & 1) + e];
          pf[H][s] = pack8v<DT>(t);
          pin_out(pf[H][s]);
        }
      });
      pin_out(psum[H]);
      pin_out(pend[H]);
    }
    if constexpr (k == 31) {
      psum[H] += pend[H];
      l_run[H] = l_run[H] * alpha[H] + psum[H];
      pin_out(l_run[H]);
      if (__builtin_amdgcn_ballot_w64(need[H]) != 0) {  // wave-uniform, rare after the first tiles
        // one d-block at a time between opaque touches of the AccVGPR tuple: the copies out of and back into the accumulator file stay
        // inside this branch (left to itself hipcc reads all 64 registers at the top of the tile loop and spills the Q fragments for it)
        static_for<DBLK>([&](auto I_) {
          constexpr int i = decltype(I_)::value;
          asm volatile("" : "+a"(o[H][i]));
          f32x16_t tv = o[H][i];
          asm volatile("" : "+v"(tv));
#pragma unroll
          for (int r = 0; r < 16; ++r) tv[r] *= alpha[H];
          asm volatile("" : "+v"(tv));
          o[H][i] = tv;
          asm volatile("s_nop 1" : "+a"(o[H][i]));
          __builtin_amdgcn_sched_barrier(0);
        });
      }
    }
  };

  // ---- S^T(half H) = K Q_H^T: 16 MFMAs, the two key blocks' chains alternate; vs(step) = the VALU slice behind each MFMA ----
  auto seg_qk = [&](auto H_, auto SAFE_, auto&& vs) {
    constexpr int H = decltype(H_)::value;
    constexpr bool SAFE = decltype(SAFE_)::value;
    asm volatile("s_nop 1" ::: "memory");
    static_for<2 * KSTEPS>([&](auto I) {
      constexpr int n = decltype(I)::value;
      constexpr int kb = n & 1, ks = n >> 1;
      mfma_sx<DT, SAFE, H == 1, ks == 0>(st[H][kb], ka[kb * KSTEPS + ks], qf[H][ks]);
      __builtin_amdgcn_sched_barrier(0);
      vs(I);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  // ---- O^T(half H) += V^T P_H^T from V slot PAR: 16 MFMAs, four d-block chains rotate; KR = 1: K fragment n of the OTHER K slot is
  //      requested behind MFMA n (the next tile's K, into the AccVGPRs this tile's S^T products are done with) ----
  auto seg_pv = [&](auto H_, auto KR_, auto SAFE_, auto&& vs) {
    constexpr int H = decltype(H_)::value, KR = decltype(KR_)::value;
    constexpr bool SAFE = decltype(SAFE_)::value;
    constexpr int SOV = 0, SOK = 0;  // the slot is part of the address registers (toggled by the tile, below)
    u32x2_t wv[2 * F4_W];
    auto issue = [&](auto F_) {  // fragment f = (d-block f & 3, k-step f >> 2)
      constexpr int f = decltype(F_)::value, i = f & 3, s = f >> 2, sl = f % F4_W;
      lds_read64_tr<SOV + (s * 16) * RB>(wv[2 * sl], av[2 * i]);
      lds_read64_tr<SOV + (s * 16 + 8) * RB>(wv[2 * sl + 1], av[2 * i + 1]);
    };
    asm volatile("s_nop 1" ::: "memory");
    static_for<F4_W>([&](auto I) { issue(I); });
    static_for<4 * DBLK>([&](auto I) {
      constexpr int f = decltype(I)::value, i = f & 3, s = f >> 2, sl = f % F4_W;
      lgkm_wait4<f4_allowed(f, KR)>();
      const u32x4_t vf = u32x4_t{wv[2 * sl][0], wv[2 * sl][1], wv[2 * sl + 1][0], wv[2 * sl + 1][1]};
      mfma_o<DT, SAFE>(o[H][i], vf, pf[H][s]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (f + F4_W < 4 * DBLK) issue(std::integral_constant<int, f + F4_W>{});
      if constexpr (KR) lds_read128_acc<SOK + (f / KSTEPS) * 32 * RB>(ka[f], ak[f % KSTEPS]);
      vs(I);
      __builtin_amdgcn_sched_barrier(0);
    });
    // the segment's last products settle HERE, in a statement without operands: whatever hipcc places in front of the next statement
    // that names an accumulator (a copy of a tuple at a control-flow merge, a spill) then reads registers the matrix pipe has written
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
    if constexpr (KR) {
      // the K fragments have landed before anything (a copy at the loop edge included) can touch them - and they are USED here: on the path
      // that leaves the tile loop nothing reads them, and a register hipcc considers dead is handed out again while the read is still in
      // flight (the first build's NaNs: late K data landing in O / Q registers)
      lgkm_wait4<0>();
      use_acc16(ka);
    }
  };

  // LDS slot of a tile = t & 1, kept IN the fragment address registers (bit 14: K slots at 0 / 0x4000, V slots at 0x8000 / 0xC000) and toggled
  // as the tile moves on, so that one tile body serves both parities: the bodies are joined by control flow, and every further body
  // costs accumulator copies and spills where the paths merge.  Before tile t: ak -> K slot of tile t + 1, av -> V slot of tile t - 1.
  auto toggle = [&](unsigned (&x)[KSTEPS]) {
#pragma unroll
    for (int i = 0; i < KSTEPS; ++i) asm volatile("v_xor_b32 %0, 0x4000, %0" : "+v"(x[i]));
  };
  auto tile = [&](int t, auto FIRST_, auto EDGE_) {
    constexpr bool FIRST = decltype(FIRST_)::value;
    using SAFE = std::integral_constant<bool, FIRST || decltype(EDGE_)::value>;
    const int kv0 = t * 64;
    // A(t)
    seg_qk(C0{}, SAFE{}, [&](auto I) {
      if constexpr (!FIRST) sm(C1{}, EDGE_, std::integral_constant<int, decltype(I)::value + 16>{}, kv0);
    });
    // B(t)
    if constexpr (FIRST) {
      settle_s(st[0][0], st[0][1]);
      static_for<16>([&](auto I) {
        sm(C0{}, EDGE_, I, kv0);
        __builtin_amdgcn_sched_barrier(0);
      });
    } else {
      seg_pv(C1{}, C0{}, SAFE{}, [&](auto I) { sm(C0{}, EDGE_, I, kv0); });
      settle_all(o);  // (see the end of the tile)
    }
    toggle(av);  // -> V slot of tile t
    // P(t): K(t+1) and V(t) (requested one tile ago) have landed for every wave; every wave is past its reads of V(t-1) (segment B) and K(t)
    // (in registers since D(t-1)): the slots V(t+1) and K(t+2) go to are free, and their copies are requested between the MFMAs of C
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // C(t)
    seg_qk(C1{}, SAFE{}, [&](auto I) {
      constexpr int k = decltype(I)::value;
      if constexpr (k == 1) stage_v(t + 1);
      if constexpr (k == 8) stage_k(t + 2);
      sm(C0{}, EDGE_, std::integral_constant<int, k + 16>{}, kv0);
    });
    // D(t)
    seg_pv(C0{}, C1{}, SAFE{}, [&](auto I) { sm(C1{}, EDGE_, I, kv0); });
    toggle(ak);  // -> K slot of tile t + 2
    // the tile bodies are joined by control flow, and where paths merge hipcc copies accumulators (v_accvgpr_mov / _read) - directly behind
    // the MFMAs it cannot see, i.e. of values the matrix pipe has not written yet (tools/check_mfma_hazards.py on the -S listing).  The
    // last products of a segment settle inside seg_pv; here every accumulator tuple is named so that such a copy can only follow.
    settle_all(o);
  };

  // ---- prologue: K(0), V(0), K(1); K(0) into registers ----
  stage_k(0);
  stage_v(0);
  if (ntiles > 1) stage_k(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  static_for<2 * KSTEPS>([&](auto I) {
    constexpr int n = decltype(I)::value;
    lds_read128_acc<(n / KSTEPS) * 32 * RB>(ka[n], ak[n % KSTEPS]);
  });
  lgkm_wait4<0>();
  toggle(ak);  // -> K slot 1 (tile 1)
  toggle(av);  // -> V slot 1 ("tile -1"; the first tile has no product segment B and toggles it back)

  // tiles [0, n_full) need no masking for any row of this block
  const int n_full = min(ntiles, CAUSAL ? min(q0, len) / 64 : len / 64);
  if (n_full > 0) tile(0, std::true_type{}, std::false_type{});
  else tile(0, std::true_type{}, std::true_type{});
  int t = 1;
  for (; t < n_full; ++t) tile(t, std::false_type{}, std::false_type{});
  for (; t < ntiles; ++t) tile(t, std::false_type{}, std::true_type{});
  // ---- drain: the second part of softmax(h1, last tile), then its product ----
  static_for<16>([&](auto I) {
    sm(C1{}, std::false_type{}, std::integral_constant<int, decltype(I)::value + 16>{}, 0);
    __builtin_amdgcn_sched_barrier(0);
  });
  seg_pv(C1{}, C0{}, std::true_type{}, [&](auto) {});  // (av -> V slot of the last tile)
  settle_all(o);

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (copies requested past the last tile: nothing may land in LDS after the block has left)
  // ---- finalize ----
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int qrow = hh ? qrow1 : qrow0;
    const unsigned lu = __float_as_uint(l_run[hh]);
    const auto lr = __builtin_amdgcn_permlane32_swap(lu, lu, false, false);
    const float l_tot = __uint_as_float(lr[0]) + __uint_as_float(lr[1]);
    const bool valid = (qrow < len);
    const float inv = (valid && l_tot > 0.f) ? 1.0f / l_tot : 0.f;
    if (qrow < S) {
      uint16_t* op = a.o + ((int64_t)b * S + qrow) * a.ldo + (int64_t)h * D;
#pragma unroll
      for (int i = 0; i < DBLK; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = 32 * i + 8 * g + 4 * hi;
          *(uint2*)(op + d) = make_uint2(pack2<DT>(o[hh][i][4 * g + 0] * inv, o[hh][i][4 * g + 1] * inv),
                                         pack2<DT>(o[hh][i][4 * g + 2] * inv, o[hh][i][4 * g + 3] * inv));
        }
      if (hi == 0)
        a.lse[((int64_t)b * a.H + h) * a.S_pad + qrow] = (valid && l_tot > 0.f) ? (m_run[hh] + log2f(l_tot)) * 0.6931471805599453f : 0.f;
    }
  }
}

template <int DT, bool CAUSAL>
int launch_fwd4(const Fwd4Args& a, hipStream_t st) {
  constexpr size_t lds = 4 * 64 * 128 * 2;
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)attn_fwd4_k<DT, CAUSAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL((attn_fwd4_k<DT, CAUSAL>), dim3(xcd_grid(a.B * a.H, (a.S + 255) / 256)), dim3(256), lds, st, a);
  MH_LAUNCH_CHECK();
}

}  // namespace

int launch_attn_fwd_wave64(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo, float* lse,
                           const int32_t* seqlens, int B, int S, int H, int causal, int dt, hipStream_t st) {
  Fwd4Args a;
  a.q = (const uint16_t*)q; a.k = (const uint16_t*)k; a.v = (const uint16_t*)v; a.o = (uint16_t*)o;
  a.lse = lse; a.seqlens = seqlens; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.B = B; a.S = S; a.H = H; a.S_pad = (S + 63) / 64 * 64;
  a.scale_log2 = (1.0f / sqrtf(128.0f)) * 1.4426950408889634f;
  if (dt == MH_BF16) return causal ? launch_fwd4<MH_BF16, true>(a, st) : launch_fwd4<MH_BF16, false>(a, st);
  if (dt == MH_F16) return causal ? launch_fwd4<MH_F16, true>(a, st) : launch_fwd4<MH_F16, false>(a, st);
  return MH_ERR_DTYPE;
}

}  // namespace mhattn
