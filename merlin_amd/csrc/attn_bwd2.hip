// Flash attention backward v2 (gfx950): dQ, dK, dV with row-major operand tiles only.
//
// Same mathematics and MFMA orientations as attn_bwd.hip / attn_bwd_kv.hip (three deterministic kernels, the
// score-like accumulator is fed straight back as the B operand of the next product), but every transposed
// operand (K^T for dQ, dO^T for dV, Q^T for dK) is produced by LDS transpose-reads of the SAME row-major tile that
// feeds the score product (attn_tiles.h): no mh_attn_prep_v passes, no Q^T/dO^T/K^T copies, no workspace, one
// third less global->LDS traffic.  Tiles are 64 rows (two 32-row halves per barrier), all LDS fragment reads are
// hand-waited inline asm in rolling windows, and the tile loop is split into branch-free full tiles and edge
// tiles (causal diagonal / sequence end).
//
//   dq2  (one block per 128 queries; Q, dO fragments in registers; streams K, V tiles):
//        S^T = K Q^T, dP^T = V dO^T, dS^T = P^T o (dP^T - delta[q]);   dQ^T[d,q] += K^T dS^T    (K^T: transpose-read)
//   kv2<1> dV (one block per 128 keys; K fragments in registers; streams Q, dO tiles):
//        S = Q K^T -> P;                                               dV^T[d,kv] += dO^T P      (dO^T: transpose-read)
//   kv2<2> dK (K, V fragments in registers):
//        S, dP = dO V^T, dS = P o (dP - delta[q]);                     dK^T[d,kv] += Q^T dS      (Q^T: transpose-read)
// delta2_k writes -delta[q] = -rowsum(dO o O) and -lse[q]/scale: the score and dP accumulators START from them, so
// P = exp2(scale*log2e * acc_S) and dS = P o acc_dP need no per-element subtraction and (dK) no registers to hold
// lse/delta next to the accumulators.  The softmax scale of dQ/dK is applied in the epilogues.
#include "attn_tiles.h"

namespace mhattn {
#ifdef MH_KV_TIMING
int g_attn_wide_stores = 1;  // (stand-alone probe build of this file)
#else
extern int g_attn_wide_stores;  // attn_fwd2.hip (mh_attn_wide_stores)
#endif
namespace {

struct Bwd2Args {
  const uint16_t *q, *k, *v, *o, *dout;
  const float* lse;
  float* delta;       // [2][B, H, S_pad]: -delta, then lse2 = -lse/scale (both are accumulator INITIAL values)
  const float* lse2;
  uint16_t *dq, *dk, *dv;
  const int32_t* seqlens;
  int64_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  int B, S, H, S_pad;
  float scale, scale_log2, inv_scale;
  const float2* rope;  // optional: inverse RoPE (rotate-half) of dQ and dK rows at their sequence position, fused into the epilogues
  int wide;            // dq / dk / dv rows are 16-byte aligned: 16-byte epilogue stores (attn_tiles.h, store_row_wide)
  uint16_t* ds;        // dS spill (5-product backward, attn_bwd3_*): [B * H][units][16 KiB], see ds_unit(); null = the 7-product form
#ifdef MH_KV_TIMING
  unsigned long long* dbg;  // development build (tools/probes/kv_timing.py): [block][wave][8] cycles per tile segment of the fused dK|dV kernel
#endif
};
#ifdef MH_KV_TIMING
// s_memtime returns through lgkmcnt: the stamps sit only where the tile's own LDS reads have all been consumed
#define KV_STAMP(i) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); tacc[i] += t_ - tprev; tprev = t_; }
#else
#define KV_STAMP(i)
#endif

template <int N>
__device__ __forceinline__ void lgkm_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// D/8 lanes per (token, head) row: one 16-byte load of O and of dO per lane, shuffle reduction within the lane group
template <int DT, int D>
__global__ __launch_bounds__(256) void delta2_k(Bwd2Args a) {
  constexpr int LPR = D / 8, RPB = 256 / LPR;  // lanes per row, rows per block
  const int part = threadIdx.x % LPR;
  const int64_t idx = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR;  // over B*S*H
  const bool ok = idx < (int64_t)a.B * a.S * a.H;
  const int64_t ic = ok ? idx : 0;
  const int h = (int)(ic % a.H);
  const int64_t t = ic / a.H;
  const int b = (int)(t / a.S), s = (int)(t % a.S);
  float x[8], y[8];
  unpack8<DT>(*(const uint4*)(a.o + t * a.ldo + (int64_t)h * D + part * 8), x);
  unpack8<DT>(*(const uint4*)(a.dout + t * a.lddo + (int64_t)h * D + part * 8), y);
  float acc = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) acc = fmaf(x[e], y[e], acc);
#pragma unroll
  for (int o2 = LPR / 2; o2 > 0; o2 >>= 1) acc += __shfl_xor(acc, o2, 64);
  if (ok && part == 0) {
    const int64_t i = ((int64_t)b * a.H + h) * a.S_pad + s;
    a.delta[i] = -acc;
    ((float*)a.lse2)[i] = -a.lse[i] * a.inv_scale;
  }
}

// rolling-window helpers ------------------------------------------------------------------------------------
// NF row fragments (A operands) of the 32-row half at byte offset BASE of a tile: fragment n <-> k-step n; consumer(n, frag)
template <int BASE, int NF, int WMAX = 8, typename F>
__device__ __forceinline__ void stream_row_frags(const unsigned* addr, F&& consume) {
  constexpr int W = NF < WMAX ? NF : WMAX;
  u32x4_t w[W];
  static_for<W>([&](auto I) { constexpr int n = decltype(I)::value; lds_read128<BASE>(w[n], addr[n]); });
  static_for<NF>([&](auto I) {
    constexpr int n = decltype(I)::value;
    constexpr int left = NF - 1 - n;
    lgkm_wait<(left < W - 1 ? left : W - 1)>();
    consume(I, w[n % W]);
    if constexpr (n + W < NF) lds_read128<BASE>(w[n % W], addr[n + W]);
  });
}
// transposed fragments of the 32-row half starting at tile row R0: fragment f = (d-block f/2, k-step f%2), NF = 2*DBLK
// (XO: extra byte offset in the immediates - the stage of a double-buffered tile when the addresses are those of stage 0)
template <int RB, int R0, int NF, int XO = 0, typename F>
__device__ __forceinline__ void stream_tr_frags(const unsigned* addr, F&& consume) {
  u32x2_t w[8];  // window of 4 fragments
  constexpr int W = NF < 4 ? NF : 4;
  static_for<W>([&](auto I) {
    constexpr int f = decltype(I)::value;
    lds_read64_tr<XO + (R0 + (f % 2) * 16) * RB>(w[2 * f], addr[2 * (f / 2)]);
    lds_read64_tr<XO + (R0 + (f % 2) * 16 + 8) * RB>(w[2 * f + 1], addr[2 * (f / 2) + 1]);
  });
  static_for<NF>([&](auto I) {
    constexpr int f = decltype(I)::value;
    constexpr int left = NF - 1 - f;
    lgkm_wait<2 * (left < W - 1 ? left : W - 1)>();
    const u32x4_t fr = u32x4_t{w[2 * (f % W)][0], w[2 * (f % W)][1], w[2 * (f % W) + 1][0], w[2 * (f % W) + 1][1]};
    consume(I, fr);
    if constexpr (f + W < NF) {
      constexpr int g = f + W;
      lds_read64_tr<XO + (R0 + (g % 2) * 16) * RB>(w[2 * (f % W)], addr[2 * (g / 2)]);
      lds_read64_tr<XO + (R0 + (g % 2) * 16 + 8) * RB>(w[2 * (f % W) + 1], addr[2 * (g / 2) + 1]);
    }
  });
}

// MFMA whose accumulator is pinned in AccVGPRs (inline asm): for accumulators nothing but MFMAs touches until the epilogue.  The
// fused dK + dV kernel needs ~400 registers; left to itself hipcc keeps the score / dP accumulators in AccVGPRs instead and moves
// every element in and out around the softmax arithmetic (128 of the 330 VALU instructions of a tile).
template <int DT>
__device__ __forceinline__ void mfma32a(f32x16_t& acc, const u32x4_t& x, const u32x4_t& y) {
  if constexpr (DT == MH_BF16)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(x), "v"(y));
  else
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(x), "v"(y));
}
// MFMA with the accumulator in VGPRs and the B operand in AccVGPRs (the wave's resident K / V fragments).  hipcc cannot see that
// these asm statements are MFMAs, so the wait states it would insert are written out: mfma_settle() between the last MFMA of a chain
// and the first VALU read of its accumulator (19 wait states cover the longest XDL write -> VALU read rule), mfma_ready() between
// the VALU / LDS writes that initialise an accumulator and the first MFMA that reads it (2).
template <int DT>
__device__ __forceinline__ void mfma32va(f32x16_t& acc, const u32x4_t& x, const u32x4_t& y) {
  if constexpr (DT == MH_BF16)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "a"(y));
  else
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "a"(y));
}
__device__ __forceinline__ void mfma_settle(f32x16_t& a, f32x16_t& b) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void mfma_ready(f32x16_t& a, f32x16_t& b) { asm volatile("s_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void mfma_settle_acc(f32x16_t& a) { asm volatile("s_nop 15\n\ts_nop 3" : "+a"(a)); }
// 16 bytes from global memory straight into AccVGPRs (waited for by the caller's s_waitcnt vmcnt)
template <int OFF>
__device__ __forceinline__ void gload128_acc(u32x4_t& d, const void* ptr) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(d) : "v"(ptr), "n"(OFF) : "memory");
}

// Two row-fragment streams interleaved (fragment n: tile at +OFF0 for even n, +OFF1 for odd n, k-step n / 2): two independent MFMA
// chains alternate, so neither waits for its own previous accumulate.  consume(n, frag) may also carry a slice of VALU work.
#ifndef MH_SP_WINDOW
#define MH_SP_WINDOW 6  // + the 8 lse / delta reads issued just before the stream: 14 LDS requests in flight (the lgkm counter holds 15; measured 6 < 7 ~ 8)
#endif
template <int OFF0, int OFF1, int KS, typename F>
__device__ __forceinline__ void stream_row_frags2(const unsigned* addr, F&& consume) {
  constexpr int NF = 2 * KS, W = NF < MH_SP_WINDOW ? NF : MH_SP_WINDOW;
  u32x4_t w[W];
  static_for<W>([&](auto I) { constexpr int n = decltype(I)::value; lds_read128<(n % 2) ? OFF1 : OFF0>(w[n], addr[n / 2]); });
  static_for<NF>([&](auto I) {
    constexpr int n = decltype(I)::value;
    constexpr int left = NF - 1 - n;
    lgkm_wait<(left < W - 1 ? left : W - 1)>();
    consume(I, w[n % W]);
    if constexpr (n + W < NF) lds_read128<((n + W) % 2) ? OFF1 : OFF0>(w[n % W], addr[(n + W) / 2]);
  });
}
// The same for transposed fragments: stream element g = (tile g % 2, fragment g / 2), fragment f = (d-block f / 2, k-step f % 2).
template <int RB, int R0, int OFF0, int OFF1, int NFH, typename F>
__device__ __forceinline__ void stream_tr_frags2(const unsigned* addr, F&& consume) {
#ifndef MH_TR_WINDOW
#define MH_TR_WINDOW 4  // fragments in flight (2 reads each; the lgkm counter holds 15)
#endif
  constexpr int NF = 2 * NFH, W = MH_TR_WINDOW;
  u32x2_t w[2 * W];
  auto issue = [&](auto G, auto SLOT) {
    constexpr int g = decltype(G)::value, sl = decltype(SLOT)::value, f = g / 2, off = (g % 2) ? OFF1 : OFF0;
    lds_read64_tr<(R0 + (f % 2) * 16) * RB + off>(w[2 * sl], addr[2 * (f / 2)]);
    lds_read64_tr<(R0 + (f % 2) * 16 + 8) * RB + off>(w[2 * sl + 1], addr[2 * (f / 2) + 1]);
  };
  static_for<W>([&](auto I) { issue(I, I); });
  static_for<NF>([&](auto I) {
    constexpr int g = decltype(I)::value;
    constexpr int left = NF - 1 - g;
    lgkm_wait<2 * (left < W - 1 ? left : W - 1)>();
    const u32x4_t fr = u32x4_t{w[2 * (g % W)][0], w[2 * (g % W)][1], w[2 * (g % W) + 1][0], w[2 * (g % W) + 1][1]};
    consume(I, fr);
    if constexpr (g + W < NF) issue(std::integral_constant<int, g + W>{}, std::integral_constant<int, g % W>{});
  });
}

// Inverse RoPE of one gradient row held in the epilogue layout (lane: channels 32*i + 8*g + 4*hi + e of its row): the
// partner of channel d < D/2 is d + D/2 = accumulator block i + DBLK/2 of the SAME lane, so the rotation is in registers.
// tab points at the (cos, sin) row of this lane's sequence position.  Gradient of y = rope(x): x_bar = rope^T(y_bar) =
// rotation by -theta.
template <int D>
__device__ __forceinline__ void unrope_rows(f32x16_t (&v)[D / 32], const float2* tab, int hi) {
  constexpr int HB = D / 64;  // accumulator blocks per half head
#pragma unroll
  for (int i = 0; i < HB; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4* t4 = (const float4*)(tab + 32 * i + 8 * g + 4 * hi);
      const float4 t01 = t4[0], t23 = t4[1];  // (c0, s0, c1, s1), (c2, s2, c3, s3)
      const float cs[4] = {t01.x, t01.z, t23.x, t23.z}, sn[4] = {t01.y, t01.w, t23.y, t23.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float lo, hi_;
        rope_rot(v[i][4 * g + e], v[i + HB][4 * g + e], cs[e], -sn[e], lo, hi_);
        v[i][4 * g + e] = lo;
        v[i + HB][4 * g + e] = hi_;
      }
    }
}

// ------------------------------------------------------------------------------------------------------------
// dQ
// ------------------------------------------------------------------------------------------------------------
template <int DT, int D, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd2_dq_k(Bwd2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RB = D * 2, T_BYTES = 64 * RB, STAGE = 2 * T_BYTES;  // K, V tiles of 64 keys
  constexpr int KSTEPS = D / 16, DBLK = D / 32;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int nq = (a.S + 127) / 128;
  int bh_, qi;
  if (!xcd_work(a.B * a.H, nq, bh_, qi)) return;
  const int qblk = CAUSAL ? nq - 1 - qi : qi;
  const int h = bh_ % a.H, b = bh_ / a.H;
  const int S = a.S;
  const int len = a.seqlens ? min(a.seqlens[b], S) : S;
  const int q0 = qblk * 128, qw0 = q0 + wave * 32, qrow = qw0 + l31;
  uint16_t* dqp = a.dq + ((int64_t)b * S + qrow) * a.lddq + (int64_t)h * D;
  if (q0 >= len) {
    if (qrow < S)
      for (int d = hi * (D / 2); d < (hi + 1) * (D / 2); d += 4) *(uint2*)(dqp + d) = make_uint2(0, 0);
    return;
  }
  u32x4_t qf[KSTEPS], dof[KSTEPS];
  {
    const int qr = min(qrow, S - 1);
    const uint16_t* qp = a.q + ((int64_t)b * S + qr) * a.ldq + (int64_t)h * D + 8 * hi;
    const uint16_t* dp = a.dout + ((int64_t)b * S + qr) * a.lddo + (int64_t)h * D + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      qf[ks] = *(const u32x4_t*)(qp + 16 * ks);
      dof[ks] = *(const u32x4_t*)(dp + 16 * ks);
    }
  }
  const int64_t bh = (int64_t)b * a.H + h;
  const float nls = a.lse2[bh * a.S_pad + min(qrow, S - 1)];   // -lse/scale
  const float ndl = a.delta[bh * a.S_pad + min(qrow, S - 1)];  // -delta
  const float nlb = nls * a.scale_log2;                          // -lse * log2(e)
  const int kv_end = CAUSAL ? min(len, q0 + 128) : len;
  const int ntiles = (kv_end + 63) / 64;
  const uint16_t* kbase = a.k + (int64_t)b * S * a.ldk + (int64_t)h * D;
  const uint16_t* vbase = a.v + (int64_t)b * S * a.ldv + (int64_t)h * D;
  const unsigned lds0 = lds_addr_of(smem);
  const auto src_k = row_src<D>(kbase, a.ldk, S, tid), src_v = row_src<D>(vbase, a.ldv, S, tid);
  auto stage = [&](int s, int kv0) {  // scalar addressing only (attn_tiles.h, stage_rows_buf); rows >= S arrive as zeros and are masked
    const unsigned base = lds0 + (unsigned)s * STAGE + (unsigned)wave * 1024u;
    stage_rows_buf<D, 64>(src_k, kv0, base);
    stage_rows_buf<D, 64>(src_v, kv0, base + T_BYTES);
  };
  f32x16_t dqacc[DBLK];
#pragma unroll
  for (int i = 0; i < DBLK; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqacc[i][r] = 0.f;
  unsigned off_r[KSTEPS], off_t[KSTEPS];
  row_frag_offsets<D>(l31, hi, off_r);
  tr_frag_offsets<D>(lane, off_t);
  const float sc = a.scale_log2;

  // (The forward's unrolled-by-two tile loop with the stage in the ds_read immediates does not fit here: 256 registers and spills of the Q / dO
  // fragments inside the loop.  The V tile's addresses are the K tile's + T_BYTES in the immediates: 16 address adds per tile instead of 24.)
  static_assert(T_BYTES + 32 * RB < 65536, "fragment offsets must fit the 16-bit ds_read immediate");
  auto tile = [&](int j, auto EDGE_) {
    constexpr bool EDGE = decltype(EDGE_)::value;
    constexpr int SO = 0;
    const int kv0 = j * 64;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (j + 1 < ntiles) stage((j + 1) & 1, kv0 + 64);
    const unsigned sb = lds0 + (unsigned)(j & 1) * STAGE;
    unsigned ak[KSTEPS], at[KSTEPS];
#pragma unroll
    for (int i = 0; i < KSTEPS; ++i) {
      ak[i] = sb + off_r[i];
      at[i] = sb + off_t[i];
    }
    static_for<2>([&](auto HALF) {
      constexpr int hf = decltype(HALF)::value;
      const int kvh = kv0 + 32 * hf;
      if constexpr (EDGE) {
        if ((CAUSAL && kvh > qw0 + 31) || kvh >= len) return;  // nothing visible to this wave (wave-uniform)
      }
      // Both chains start from the MFMA's inline-constant zero C operand (no accumulator to initialise: the -lse / -delta splats of the
      // round-3 form were 32 v_mov per half); the row constants enter in the element work instead: p = exp2(s * sc - lse * log2e) as one fma
      // per score, dS = p * (dP - delta).
      f32x16_t sacc, pacc;
      const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      stream_row_frags<SO + hf * 32 * RB, KSTEPS>(ak, [&](auto I, const u32x4_t& fr) {
        constexpr int n = decltype(I)::value;
        if constexpr (n == 0) sacc = mfma32v<DT>(fr, qf[0], zero);
        else sacc = mfma32v<DT>(fr, qf[n], sacc);
      });
      stream_row_frags<SO + T_BYTES + hf * 32 * RB, KSTEPS>(ak, [&](auto I, const u32x4_t& fr) {
        constexpr int n = decltype(I)::value;
        if constexpr (n == 0) pacc = mfma32v<DT>(fr, dof[0], zero);
        else pacc = mfma32v<DT>(fr, dof[n], pacc);
      });
      float dsv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) dsv[r] = fast_exp2(fmaf(sacc[r], sc, nlb));
      if (EDGE && ((kvh + 32 > len) || (qw0 + 32 > len) || (CAUSAL && (kvh + 31 > qw0)))) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kvh + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const bool ok = (kv < len) && (qrow < len) && (!CAUSAL || kv <= qrow);
          dsv[r] = ok ? dsv[r] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) dsv[r] *= pacc[r] + ndl;
      const u32x4_t dsf[2] = {pack8v<DT>(dsv), pack8v<DT>(dsv + 8)};
      stream_tr_frags<RB, hf * 32, 2 * DBLK, SO>(at, [&](auto I, const u32x4_t& fr) {
        constexpr int f = decltype(I)::value;
        dqacc[f / 2] = mfma32v<DT>(fr, dsf[f % 2], dqacc[f / 2]);
      });
    });
  };
  const int n_full = min(ntiles, (CAUSAL ? min(q0, len) : len) / 64);  // and all rows of the block < len? checked per EDGE
  const bool rows_full = (q0 + 128 <= len);
  stage(0, 0);
  if (rows_full) {
    for (int j = 0; j < n_full; ++j) tile(j, std::false_type{});
    for (int j = n_full; j < ntiles; ++j) tile(j, std::true_type{});
  } else {
    for (int j = 0; j < ntiles; ++j) tile(j, std::true_type{});
  }

  if (qrow < S) {
    const bool valid = qrow < len;
#pragma unroll
    for (int i = 0; i < DBLK; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) dqacc[i][r] *= a.scale;
    if (a.rope && valid) unrope_rows<D>(dqacc, a.rope + (int64_t)qrow * (D / 2), hi);
    auto val = [&](int i, int r) { return dqacc[i][r]; };
    if (a.wide) store_row_wide<DT, DBLK>(dqp, hi, valid, val);
    else store_row_narrow<DT, DBLK>(dqp, hi, valid, val);
  }
}

// ------------------------------------------------------------------------------------------------------------
// dV (MODE 1) / dK (MODE 2)
// ------------------------------------------------------------------------------------------------------------
template <int DT, int D, bool CAUSAL, int MODE>
__global__ __launch_bounds__(256, MODE == 3 ? 1 : 2) void attn_bwd2_kv_k(Bwd2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // MODE 1: dV, 2: dK (two waves per SIMD each), 3: both from ONE score / dP computation per tile (one wave per SIMD: the two
  // accumulator sets, K and V fragments are ~200 registers before anything else)
  constexpr bool DO_DK = (MODE & 2) != 0, DO_DV = (MODE & 1) != 0;
  constexpr bool ASM = (MODE == 3);  // register classes by hand: dK / dV accumulators and K / V fragments in AccVGPRs, scores / dP in VGPRs
  constexpr int RB = D * 2, T_BYTES = 64 * RB;          // Q, dO tiles of 64 queries
  constexpr int OFF_DO = T_BYTES, OFF_LSE = 2 * T_BYTES;  // + wave*512: [lse2 64 f32 | delta 64 f32]
  constexpr int STAGE = 2 * T_BYTES + 2048;
  constexpr int KSTEPS = D / 16, DBLK = D / 32;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  int bh_, kvblk;
  if (!xcd_work(a.B * a.H, (a.S + 127) / 128, bh_, kvblk)) return;
  const int h = bh_ % a.H, b = bh_ / a.H;
  const int S = a.S;
  const int len = a.seqlens ? min(a.seqlens[b], S) : S;
  const int kv0 = kvblk * 128, kvw0 = kv0 + wave * 32, kvrow = kvw0 + l31;
  uint16_t* outk = a.dk + ((int64_t)b * S + kvrow) * a.lddk + (int64_t)h * D;
  uint16_t* outv = a.dv + ((int64_t)b * S + kvrow) * a.lddv + (int64_t)h * D;
  if (kv0 >= len) {
    if (kvrow < S)
      for (int d = hi * (D / 2); d < (hi + 1) * (D / 2); d += 4) {
        if constexpr (DO_DK) *(uint2*)(outk + d) = make_uint2(0, 0);
        if constexpr (DO_DV) *(uint2*)(outv + d) = make_uint2(0, 0);
      }
    return;
  }
  // K (and V) fragments of this wave's 32 keys: B operands
  u32x4_t kf[KSTEPS], vf[DO_DK ? KSTEPS : 1];
  {
    const int kr = min(kvrow, S - 1);
    const uint16_t* kp = a.k + ((int64_t)b * S + kr) * a.ldk + (int64_t)h * D + 8 * hi;
    const uint16_t* vp = a.v + ((int64_t)b * S + kr) * a.ldv + (int64_t)h * D + 8 * hi;
    if constexpr (ASM) {  // into AccVGPRs: read only as MFMA B operands (the first tile's s_waitcnt vmcnt(0) covers them)
      static_for<KSTEPS>([&](auto I) {
        constexpr int ks = decltype(I)::value;
        gload128_acc<32 * ks>(kf[ks], kp);
        gload128_acc<32 * ks>(vf[ks], vp);
      });
    } else {
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        kf[ks] = *(const u32x4_t*)(kp + 16 * ks);
        if constexpr (DO_DK) vf[ks] = *(const u32x4_t*)(vp + 16 * ks);
      }
    }
  }
  const int q_begin = CAUSAL ? kv0 : 0;
  const int ntiles = (len - q_begin + 63) / 64;
  const uint16_t* qbase = a.q + (int64_t)b * S * a.ldq + (int64_t)h * D;
  const uint16_t* dobase = a.dout + (int64_t)b * S * a.lddo + (int64_t)h * D;
  const float* lse_row = a.lse2 + ((int64_t)b * a.H + h) * a.S_pad;
  const float* dl_row = a.delta + ((int64_t)b * a.H + h) * a.S_pad;
  const auto src_q = row_src<D>(qbase, a.ldq, S, tid), src_do = row_src<D>(dobase, a.lddo, S, tid);
  const unsigned lds_stage0 = lds_addr_of(smem) + (unsigned)wave * 1024u;
  auto stage = [&](int s, int q0) {
    char* base = smem + s * STAGE;
    stage_rows_buf<D, 64>(src_q, q0, lds_stage0 + (unsigned)s * STAGE);  // scalar addressing only; rows >= S arrive as zeros and are masked
    stage_rows_buf<D, 64>(src_do, q0, lds_stage0 + (unsigned)s * STAGE + OFF_DO);
    // per-wave copy: lanes 0-63 -> lse2[q0 + lane], then delta[q0 + lane] (2 x 256 contiguous LDS bytes)
    glds4(lse_row + q0 + lane, base + OFF_LSE + wave * 512);
    glds4(dl_row + q0 + lane, base + OFF_LSE + wave * 512 + 256);
  };
  f32x16_t acck[DO_DK ? DBLK : 1], accv[DO_DV ? DBLK : 1];
#pragma unroll
  for (int i = 0; i < DBLK; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if constexpr (DO_DK) acck[i][r] = 0.f;
      if constexpr (DO_DV) accv[i][r] = 0.f;
    }
  const unsigned lds0 = lds_addr_of(smem);
  unsigned off_r[KSTEPS], off_t[KSTEPS];
  row_frag_offsets<D>(l31, hi, off_r);
  tr_frag_offsets<D>(lane, off_t);
  const unsigned off_l = OFF_LSE + wave * 512 + hi * 16;
  const float sc = a.scale_log2;
#ifdef MH_KV_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#endif

  auto tile = [&](int j, auto EDGE_) {
    constexpr bool EDGE = decltype(EDGE_)::value;
    const int q0 = q_begin + j * 64;
    if constexpr (MODE == 3 && !EDGE) { KV_STAMP(0); }  // [0] the previous tile's last segment (D), or everything before the first full tile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MODE == 3 && !EDGE) { KV_STAMP(1); }  // [1] vmcnt(0) + barrier
    if (j + 1 < ntiles) stage((j + 1) & 1, q0 + 64);
    const unsigned sb = lds0 + (unsigned)(j & 1) * STAGE;
    unsigned aq[KSTEPS], ado[KSTEPS], atq[DO_DK ? KSTEPS : 1], atdo[DO_DV ? KSTEPS : 1];
#pragma unroll
    for (int i = 0; i < KSTEPS; ++i) {
      aq[i] = sb + off_r[i];
      ado[i] = sb + OFF_DO + off_r[i];
      if constexpr (DO_DK) atq[i] = sb + off_t[i];            // dK: Q^T from the Q tile
      if constexpr (DO_DV) atdo[i] = sb + OFF_DO + off_t[i];  // dV: dO^T from the dO tile
    }
    const unsigned al = sb + off_l;
    if constexpr (MODE == 3 && !EDGE) {
      // Full tile, one wave per SIMD: nothing else runs its VALU beside this wave's MFMAs, so the two 32-query halves are software-
      // pipelined by hand.  A: S, dP of half 0 (two interleaved accumulate chains).  B: S, dP of half 1, with P = exp2(.) and
      // dS = P o dP of half 0 sliced between its MFMAs.  C: dV, dK of half 0 (interleaved), with the element work of half 1 in
      // between.  D: dV, dK of half 1.  Same products, operands and accumulation order as the sequential form: bit-identical.
      f32x16_t sa[2], pa[2];
      float pv[2][16], dsv[2][16];
      u32x4_t pf[2][2], dsf[2][2];
      auto sp = [&](auto HALF, auto&& slice) {
        constexpr int hf = decltype(HALF)::value;
        u32x4_t lsev[4], dlv[4];
        static_for<4>([&](auto I) { constexpr int g = decltype(I)::value; lds_read128<hf * 128 + 32 * g>(lsev[g], al); });
        static_for<4>([&](auto I) { constexpr int g = decltype(I)::value; lds_read128<256 + hf * 128 + 32 * g>(dlv[g], al); });
        stream_row_frags2<hf * 32 * RB, hf * 32 * RB + OFF_DO, KSTEPS>(aq, [&](auto I, const u32x4_t& fr) {
          constexpr int n = decltype(I)::value;
          if constexpr (n == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                sa[hf][4 * g + e] = __uint_as_float(lsev[g][e]);
                pa[hf][4 * g + e] = __uint_as_float(dlv[g][e]);
              }
          }
          if constexpr (n == 0) mfma_ready(sa[hf], pa[hf]);
          if constexpr (n % 2 == 0) mfma32va<DT>(sa[hf], fr, kf[n / 2]);
          else mfma32va<DT>(pa[hf], fr, vf[n / 2]);
          slice(I);
        });
        mfma_settle(sa[hf], pa[hf]);
      };
      constexpr int EPS = 16 / (2 * KSTEPS);  // elements per slice: both streams have 2 * KSTEPS = 4 * DBLK fragments
      static_assert(EPS * 2 * KSTEPS == 16 && 4 * DBLK == 2 * KSTEPS, "slices cover the 16 accumulator elements");
      auto elem = [&](auto HALF, auto I) {  // slice I of half HALF: P and dS of EPS elements
        constexpr int hf = decltype(HALF)::value;
#pragma unroll
        for (int r = decltype(I)::value * EPS; r < (decltype(I)::value + 1) * EPS; ++r) {
          pv[hf][r] = fast_exp2(sa[hf][r] * sc);
          dsv[hf][r] = pv[hf][r] * pa[hf][r];
        }
      };
      auto packs = [&](auto HALF) {
        constexpr int hf = decltype(HALF)::value;
        pf[hf][0] = pack8v<DT>(pv[hf]); pf[hf][1] = pack8v<DT>(pv[hf] + 8);
        dsf[hf][0] = pack8v<DT>(dsv[hf]); dsf[hf][1] = pack8v<DT>(dsv[hf] + 8);
      };
      auto dvdk = [&](auto HALF, auto&& slice) {
        constexpr int hf = decltype(HALF)::value;
        stream_tr_frags2<RB, hf * 32, OFF_DO, 0, 2 * DBLK>(atq, [&](auto I, const u32x4_t& fr) {
          constexpr int g = decltype(I)::value, f = g / 2;
          if constexpr (g % 2 == 0) mfma32a<DT>(accv[f / 2], fr, pf[hf][f % 2]);
          else mfma32a<DT>(acck[f / 2], fr, dsf[hf][f % 2]);
          slice(I);
        });
      };
      using H0 = std::integral_constant<int, 0>;
      using H1 = std::integral_constant<int, 1>;
      KV_STAMP(2);  // [2] copy requests of the next tile + fragment addresses
      sp(H0{}, [&](auto) {});
      KV_STAMP(3);  // [3] segment A: S, dP of half 0
      sp(H1{}, [&](auto I) { elem(H0{}, I); });
      packs(H0{});
      KV_STAMP(4);  // [4] segment B
      dvdk(H0{}, [&](auto I) { elem(H1{}, I); });
      packs(H1{});
      KV_STAMP(5);  // [5] segment C
      dvdk(H1{}, [&](auto) {});
#ifdef MH_KV_TIMING
      tacc[6] += 1;  // full tiles
#endif
      return;
    }
    static_for<2>([&](auto HALF) {
      constexpr int hf = decltype(HALF)::value;
      const int qh = q0 + 32 * hf;
      if constexpr (EDGE) {
        if ((CAUSAL && kvw0 > qh + 31) || qh >= len) return;  // this wave's keys see none of these queries
      }
      // -lse/scale and -delta of the 16 query rows this lane's accumulator registers stand for (q = qh + 8*g + 4*hi
      // + e) are read straight into the accumulators: issued ahead of the fragment stream (LDS returns in order, so the
      // stream's first wait covers them) and moved in just before the first MFMA.
      f32x16_t sacc, pacc;
      u32x4_t lsev[4];
      static_for<4>([&](auto I) { constexpr int g = decltype(I)::value; lds_read128<hf * 128 + 32 * g>(lsev[g], al); });
      stream_row_frags<hf * 32 * RB, KSTEPS>(aq, [&](auto I, const u32x4_t& fr) {
        if constexpr (decltype(I)::value == 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) sacc[4 * g + e] = __uint_as_float(lsev[g][e]);
        }
        if constexpr (ASM) {
          if constexpr (decltype(I)::value == 0) mfma_ready(sacc, sacc);
          mfma32va<DT>(sacc, fr, kf[decltype(I)::value]);
        } else {
          sacc = mfma32v<DT>(fr, kf[decltype(I)::value], sacc);
        }
      });
      if constexpr (DO_DK) {
        u32x4_t dlv[4];
        static_for<4>([&](auto I) { constexpr int g = decltype(I)::value; lds_read128<256 + hf * 128 + 32 * g>(dlv[g], al); });
        stream_row_frags<hf * 32 * RB, KSTEPS, 4>(ado, [&](auto I, const u32x4_t& fr) {
          if constexpr (decltype(I)::value == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
              for (int e = 0; e < 4; ++e) pacc[4 * g + e] = __uint_as_float(dlv[g][e]);
          }
          if constexpr (ASM) {
            if constexpr (decltype(I)::value == 0) mfma_ready(pacc, pacc);
            mfma32va<DT>(pacc, fr, vf[decltype(I)::value]);
          } else {
            pacc = mfma32v<DT>(fr, vf[decltype(I)::value], pacc);
          }
        });
      }
      if constexpr (ASM) mfma_settle(sacc, pacc);
      float pv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) pv[r] = fast_exp2(sacc[r] * sc);
      if (EDGE && ((qh + 32 > len) || (kvw0 + 32 > len) || (CAUSAL && (kvw0 + 31 > qh)))) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int q = qh + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const bool ok = (q < len) && (kvrow < len) && (!CAUSAL || kvrow <= q);
          pv[r] = ok ? pv[r] : 0.f;
        }
      }
      if constexpr (DO_DV) {
        const u32x4_t pf[2] = {pack8v<DT>(pv), pack8v<DT>(pv + 8)};
        stream_tr_frags<RB, hf * 32, 2 * DBLK>(atdo, [&](auto I, const u32x4_t& fr) {
          constexpr int f = decltype(I)::value;
          if constexpr (ASM) mfma32a<DT>(accv[f / 2], fr, pf[f % 2]);
          else accv[f / 2] = mfma32v<DT>(fr, pf[f % 2], accv[f / 2]);
        });
      }
      if constexpr (DO_DK) {
#pragma unroll
        for (int r = 0; r < 16; ++r) pv[r] *= pacc[r];  // dS (unscaled)
        const u32x4_t dsf[2] = {pack8v<DT>(pv), pack8v<DT>(pv + 8)};
        stream_tr_frags<RB, hf * 32, 2 * DBLK>(atq, [&](auto I, const u32x4_t& fr) {
          constexpr int f = decltype(I)::value;
          if constexpr (ASM) mfma32a<DT>(acck[f / 2], fr, dsf[f % 2]);
          else acck[f / 2] = mfma32v<DT>(fr, dsf[f % 2], acck[f / 2]);
        });
      }
    });
  };
  // tiles: diagonal tiles first (j < n_diag), then fully visible ones, then the tail at the sequence end
  const bool keys_full = (kv0 + 128 <= len);
  const int n_diag = CAUSAL ? min(ntiles, 2) : 0;                     // q0 in {kv0, kv0+64}: touches the diagonal
  const int n_tail = (len % 64) ? 1 : 0;                               // last tile crosses `len`
  const int j_full_end = keys_full ? max(n_diag, ntiles - n_tail) : n_diag;
  stage(0, q_begin);
  for (int j = 0; j < n_diag; ++j) tile(j, std::true_type{});
#ifdef MH_KV_TIMING
  if constexpr (MODE == 3) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev) :: "memory");
#endif
  for (int j = n_diag; j < j_full_end; ++j) tile(j, std::false_type{});
  for (int j = max(n_diag, j_full_end); j < ntiles; ++j) tile(j, std::true_type{});
#ifdef MH_KV_TIMING
  if constexpr (MODE == 3) {
    KV_STAMP(7);  // [7] last full tile's segment D + trailing edge tiles (not meaningful)
    if (a.dbg && lane == 0)
      for (int i = 0; i < 8; ++i) a.dbg[((int64_t)blockIdx.x * 4 + wave) * 8 + i] = tacc[i];
  }
#endif

  if constexpr (ASM) {
#pragma unroll
    for (int i = 0; i < DBLK; ++i) { mfma_settle_acc(accv[i]); mfma_settle_acc(acck[i]); }
  }
  if (kvrow < S) {
    const bool valid = kvrow < len;
    auto store_rows = [&](auto& acc, uint16_t* outp) {
      auto val = [&](int i, int r) { return acc[i][r]; };
      if (a.wide) store_row_wide<DT, DBLK>(outp, hi, valid, val);
      else store_row_narrow<DT, DBLK>(outp, hi, valid, val);
    };
    if constexpr (DO_DV) store_rows(accv, outv);
    if constexpr (DO_DK) {
#pragma unroll
      for (int i = 0; i < DBLK; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acck[i][r] *= a.scale;
      if (a.rope && valid) unrope_rows<D>(acck, a.rope + (int64_t)kvrow * (D / 2), hi);
      store_rows(acck, outk);
    }
  }
}


// ------------------------------------------------------------------------------------------------------------
// dK | dV, round-5 form (D = 128): attn_bwd3_kv_k
// ------------------------------------------------------------------------------------------------------------
// Same products, operands and per-accumulator summation order as attn_bwd2_kv_k<MODE 3> (bit-identical dK, dV), different plumbing.  What the
// s_memtime stamps of that kernel showed (tools/probes/kv_timing.py, profiles/r05_kv_timing.txt): of ~4250 cycles per 64-query tile (64 MFMAs =
// 2048) ~900 pass at the top of the tile with the matrix pipe idle - the barrier (155) and, above all, the ten LDS-DMA copy requests of the next
// tile (750: an LDS-DMA instruction holds a lone wave's issue port ~75 cycles, and with one wave per SIMD nobody else fills it).  Here:
//   * tile copies are REGISTER-STAGED (cdna guide T14): 8 buffer_load_dwordx4 + 1 buffer_load_dwordx2 (lse | delta) per wave and tile into 34
//     VGPRs a whole tile ahead, written into LDS by 8 ds_write_b128 + 1 ds_write_b64 - every one of them a single-issue instruction behind
//     its own MFMA (a ds_write_b128 issues in ~13 cycles, a buffer_load in ~16);
//   * THREE LDS stages and ONE barrier per tile, in its MIDDLE (between the S / dP products and the dV / dK products): a wave writes tile j + 1
//     into stage (j + 1) % 3 in the first half of tile j - every wave has left tile j - 2, the stage's last reader, when it arrives at the
//     barrier of tile j - 1 - and reads it after the barrier of tile j.  Nothing is waited for at a tile seam;
//   * the stage is a template parameter of the tile body (loop unrolled by 3): every LDS address is a loop-invariant register + immediate
//     (the 33 v_add of fragment addresses per tile are gone);
//   * the contraction steps of the dV / dK products run k-step-major (all d-blocks of k-step 0, then k-step 1 - per accumulator the order is
//     unchanged), so the second half of a 32-query half's exp / dS arithmetic sits behind the first half's MFMAs.
// Tiles that need masks (the two diagonal tiles, a sequence end) and the tiles left over by the unrolling run a plain sequential body on
// the same copy / barrier protocol.
__device__ __forceinline__ void bload128(u32x4_t& d, unsigned voff, const i32x4_t& srd) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(d) : "v"(voff), "s"(srd) : "memory");
}
__device__ __forceinline__ void bload64(u32x2_t& d, unsigned voff, const i32x4_t& srd, unsigned soff) {
  asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(d) : "v"(voff), "s"(srd), "s"(soff) : "memory");
}
// 16 bytes from global memory into VGPRs; waited for by the caller's counted s_waitcnt vmcnt (hipcc would otherwise count only the loads it
// can see and drain the LDS-DMA copies queued between them)
template <int OFF>
__device__ __forceinline__ void gload128(u32x4_t& d, const void* ptr) {
#ifdef MH_SPILL_TEMPORAL
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(d) : "v"(ptr), "n"(OFF) : "memory");
#else
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2 nt" : "=v"(d) : "v"(ptr), "n"(OFF) : "memory");  // (streamed once)
#endif
}
template <int OFF>
__device__ __forceinline__ void lds_write128(unsigned addr, const u32x4_t& v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_write64(unsigned addr, const u32x2_t& v) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
namespace kv3 {
constexpr int RB = 256, T_BYTES = 64 * RB;  // D = 128: 16-KiB tiles of 64 rows
constexpr int LSE_AREA = 2048;              // per stage: 4 waves x [lse2 64 f32 | delta 64 f32]
constexpr int TILES0 = 3 * LSE_AREA;        // the three [Q tile | dO tile] stages start behind the three lse areas
constexpr int STG = 2 * T_BYTES;
constexpr int LDS_BYTES = TILES0 + 3 * STG;  // 104 448
constexpr int OFF_DO = T_BYTES;
// stage s lives at TILES0 + s * STG: address register set (s == 2), immediate (s & 1) * STG  (ds offsets are 16 bits)
constexpr int stage_set(int s) { return s == 2 ? 1 : 0; }
constexpr int stage_imm(int s) { return (s & 1) * STG; }
constexpr int W_ROW = 6, W_TR = 4;  // fragments in flight of the two stream kinds
// lgkmcnt in front of MFMA g of a fragment stream: the reads of fragments g + 1 .. g + W - 1 (RPF instructions each) and the LDS fillers of
// gaps g - W + 1 .. g - 1 (XL::at) are younger than fragment g's (fragment g + W is requested at the END of gap g, behind that gap's fillers)
template <typename XL, int W, int RPF>
constexpr int lgkm_before(int g) {
  int n = RPF * ((15 - g) < (W - 1) ? (15 - g) : (W - 1));
  for (int h = (g - W + 1 > 0 ? g - W + 1 : 0); h < g; ++h) n += XL::at(h);
  return n;
}
struct XL0 { static constexpr int at(int) { return 0; } };
// The LDS instruction order of a fast tile (attn_bwd3_kv_k, tile_fast) and the lgkmcnt in front of each of its 64 MFMAs.  Behind MFMA g, in
// this order: the gap's LDS fillers (gaps 0 .. 8 one ds_write of tile j + 1, gaps 9 .. 12 two lse / delta reads of half 1, gaps 50 .. 57 one
// lse / delta read of the next tile's half 0), then the request of transposed fragment g + 4 (two reads, when 32 <= g + 4 < 64), then of row
// fragment g + 6 (one read; of the next tile when g + 6 >= 64).  LDS instructions complete in order, so the wait in front of MFMA g is the
// number of LDS instructions issued after the youngest one it needs: its fragment's last read, and for g = 16 the last of half 1's lse reads
// (g = 0: the next tile's half-0 lse reads are older than its fragment 0).  The sequence is periodic in the tile.
struct Sched {
  static constexpr int fillers(int g) { return (g <= 8 ? 1 : 0) + ((g >= 9 && g <= 12) ? 2 : 0) + ((g >= 50 && g <= 57) ? 1 : 0); }
  static constexpr int tr_reads(int g) { return (g + 4 >= 32 && g + 4 < 64) ? 2 : 0; }       // request of transposed fragment g + 4
  static constexpr int row_reads(int g) { return (g + 6 < 32 || g + 6 >= 64) ? 1 : 0; }      // request of row fragment (g + 6) % 64
  static constexpr int in_gap(int g) { return fillers(g) + tr_reads(g) + row_reads(g); }
  // LDS instructions issued in gaps [g0, g1) of the periodic sequence (g0 <= g1, any integers)
  static constexpr int issued(int g0, int g1) {
    int n = 0;
    for (int g = g0; g < g1; ++g) n += in_gap(((g % 64) + 64) % 64);
    return n;
  }
  static constexpr int wait(int g) {
    // gap in which fragment g was requested (relative to this tile; negative = previous tile) and the instructions of that gap behind its last read
    const int rq = g < 32 ? g - 6 : g - 4;
    const int rqm = ((rq % 64) + 64) % 64;
    int after = g < 32 ? 0 : row_reads(rqm);  // a row request of the same gap sits behind the transposed one
    int n = after + issued(rq + 1, g);
    if (g == 16) {  // half 1's lse reads: the last one is the second filler of gap 12, in front of that gap's fragment request
      const int m = tr_reads(12) + row_reads(12) + issued(13, 16);
      n = m < n ? m : n;
    }
    return n;
  }
};
// segment A: gaps 0 .. 8 carry one ds_write each, gaps 8 .. 15 one lse / delta read of half 1 each
struct XLA { static constexpr int at(int g) { return (g <= 8 ? 1 : 0) + (g >= 8 ? 1 : 0); } };
}  // namespace kv3

// dS spill layout (SPILL = true; causal, S % 128 == 0, no ragged lengths): the unscaled dS = P o (dP - delta) of key block kb (128 keys) and query
// tile qt (64 queries, qt >= 2 kb) is one 16-KiB unit, units of a (batch, head) ordered by (kb, qt):  [wave 4][half 2][k-step 2][lane 64] x 16 B,
// where lane (hi, l31)'s 16 bytes are the 8 queries 64 qt + 32 half + 16 k-step + 8 hi .. + 7 of key 128 kb + 32 wave + l31 (the lane's own
// two 4-query groups after a half-wave exchange, cdna guide T21): every store instruction of a wave writes 1 KiB of contiguous memory.
// attn_bwd3_dq_k reads the units back as dQ = dS K (one product instead of the three of attn_bwd2_dq_k).
__host__ __device__ inline int64_t ds_units_per_bh(int S) { const int64_t nkb = S / 128; return nkb * nkb + nkb; }
__device__ __forceinline__ int ds_unit(int kb, int qt, int nqt) { return kb * nqt - kb * (kb - 1) + (qt - 2 * kb); }
template <int OFF>
__device__ __forceinline__ void gstore128(const void* base, unsigned voff, const u32x4_t& v) {
  asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3" ::"v"(voff), "v"(v), "s"(base), "n"(OFF) : "memory");
}

template <int DT, bool CAUSAL, bool SPILL = false>
__global__ __launch_bounds__(256, 1) void attn_bwd3_kv_k(Bwd2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using namespace kv3;
  constexpr int D = 128, KSTEPS = 8, DBLK = 4;
  using std::integral_constant;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  int bh_, kvblk;
  if (!xcd_work(a.B * a.H, (a.S + 127) / 128, bh_, kvblk)) return;
  const int h = bh_ % a.H, b = bh_ / a.H;
  const int S = a.S;
  const int len = a.seqlens ? min(a.seqlens[b], S) : S;
  const int kv0 = kvblk * 128, kvw0 = kv0 + wave * 32, kvrow = kvw0 + l31;
  uint16_t* outk = a.dk + ((int64_t)b * S + kvrow) * a.lddk + (int64_t)h * D;
  uint16_t* outv = a.dv + ((int64_t)b * S + kvrow) * a.lddv + (int64_t)h * D;
  if (kv0 >= len) {
    if (kvrow < S)
      for (int d = hi * (D / 2); d < (hi + 1) * (D / 2); d += 4) {
        *(uint2*)(outk + d) = make_uint2(0, 0);
        *(uint2*)(outv + d) = make_uint2(0, 0);
      }
    return;
  }
  // K and V fragments of this wave's 32 keys: MFMA B operands, resident in AccVGPRs (the prologue's vmcnt(0) covers them)
  u32x4_t kf[KSTEPS], vf[KSTEPS];
  {
    const int kr = min(kvrow, S - 1);
    const uint16_t* kp = a.k + ((int64_t)b * S + kr) * a.ldk + (int64_t)h * D + 8 * hi;
    const uint16_t* vp = a.v + ((int64_t)b * S + kr) * a.ldv + (int64_t)h * D + 8 * hi;
    static_for<KSTEPS>([&](auto I) {
      constexpr int ks = decltype(I)::value;
      gload128_acc<32 * ks>(kf[ks], kp);
      gload128_acc<32 * ks>(vf[ks], vp);
    });
  }
  const int q_begin = CAUSAL ? kv0 : 0;
  const int ntiles = (len - q_begin + 63) / 64;

  // ---- tile copies: global -> registers -> LDS ------------------------------------------------------------------------------------------
  // thread t of copy instruction i holds the 16-byte chunk (t % 16) ^ swz(row) of tile row 16 i + t / 16 and writes it at the lane-linear
  // position (256 i + t) * 16 of the tile: the same LDS image the LDS-DMA copies of attn_tiles.h produce (every fragment read is unchanged).
  // Rows past the end of the batch element fail the descriptor's range check and arrive as zeros (voffset takes part in it; the tile's first
  // row goes into the descriptor's base, num_records = what is left of the batch element from there).
  const uint64_t gq = (uint64_t)(uintptr_t)(a.q + (int64_t)b * S * a.ldq + (int64_t)h * D);
  const uint64_t gdo = (uint64_t)(uintptr_t)(a.dout + (int64_t)b * S * a.lddo + (int64_t)h * D);
  const unsigned rbq = (unsigned)a.ldq * 2u, rbdo = (unsigned)a.lddo * 2u;
  const unsigned spanq = (unsigned)S * rbq, spando = (unsigned)S * rbdo;
  unsigned voq[4], vodo[4];
  {
    const int row = tid >> 4, cc = tid & 15;
    const unsigned c16 = (unsigned)((cc ^ TileSwz<D>::f(row)) * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      voq[i] = (unsigned)(row + 16 * i) * rbq + c16;
      vodo[i] = (unsigned)(row + 16 * i) * rbdo + c16;
    }
  }
  // lse2 | delta: lanes 0-31 fetch lse2[q0 + 2 l .. + 1], lanes 32-63 delta[q0 + 2 l ..] (the two arrays are one allocation: delta first)
  const uint64_t gld = (uint64_t)(uintptr_t)a.delta;
  const unsigned plane = (unsigned)((int64_t)a.B * a.H * a.S_pad * 4);
  const unsigned vol = (hi ? 0u : plane) + (unsigned)(((int64_t)b * a.H + h) * a.S_pad + 2 * l31) * 4u;
  const i32x4_t srd_l = row_srd(gld, 2u * plane);
  u32x4_t rq[4], rdo[4];
  u32x2_t rl;
  struct TileSrd { i32x4_t q, d; unsigned so; };
  auto tile_srd = [&](int j) {  // descriptors of tile j (scalar arithmetic; pinned so that they are formed well ahead of the loads that read them)
    const unsigned r0 = (unsigned)(q_begin + 64 * j);
    const unsigned aq = r0 * rbq, ad = r0 * rbdo;  // (< 2^32: the launcher checks S * ld * 2 < 2^31, r0 < S + 192)
    TileSrd t;
    // (span - a as a wrapped 32-bit difference, then a SIGNED max with 0: span < 2^31 and a tile starts at most three tiles past the end, so the
    //  difference is representable; written as an unsigned compare-and-subtract hipcc forms a saturating subtract, which only exists on the
    //  VALU, and the backend then refuses the VGPR -> SGPR copy - cf. left_after() in gemm_w4.hip)
    t.q = row_srd(gq + aq, (unsigned)max((int)(spanq - aq), 0));
    t.d = row_srd(gdo + ad, (unsigned)max((int)(spando - ad), 0));
    t.so = min(r0, (unsigned)(a.S_pad - 64)) * 4u;  // (tiles past the end: any valid rows, never used)
    asm volatile("" : "+s"(t.q), "+s"(t.d), "+s"(t.so));
    return t;
  };
  auto ld_q = [&](auto I, const TileSrd& t) {
    constexpr int i = decltype(I)::value;
    bload128(rq[i], voq[i], t.q);
  };
  auto ld_do = [&](auto I, const TileSrd& t) {
    constexpr int i = decltype(I)::value;
    bload128(rdo[i], vodo[i], t.d);
  };
  auto ld_l = [&](const TileSrd& t) {
    bload64(rl, vol, srd_l, t.so);
  };
  const unsigned lds0 = lds_addr_of(smem);
  const unsigned wb[2] = {lds0 + TILES0 + (unsigned)tid * 16u, lds0 + TILES0 + 2 * STG + (unsigned)tid * 16u};  // tile writes: [set] + stage_imm + 4096 i (+ OFF_DO)
  const unsigned wl = lds0 + (unsigned)wave * 512u + (unsigned)lane * 8u;                                         // lse | delta write: + LSE_AREA * stage
  // fragment addresses: two register sets (stages 0 / 1 and stage 2), the rest is immediates
  unsigned ar[2][KSTEPS], at[2][KSTEPS];
  {
    unsigned off_r[KSTEPS], off_t[KSTEPS];
    row_frag_offsets<D>(l31, hi, off_r);
    tr_frag_offsets<D>(lane, off_t);
#pragma unroll
    for (int i = 0; i < KSTEPS; ++i) {
      ar[0][i] = lds0 + TILES0 + off_r[i]; ar[1][i] = ar[0][i] + 2 * STG;
      at[0][i] = lds0 + TILES0 + off_t[i]; at[1][i] = at[0][i] + 2 * STG;
    }
  }
  const unsigned al = lds0 + (unsigned)wave * 512u + (unsigned)hi * 16u;  // lse reads: + LSE_AREA * stage + 128 hf + 32 g (+ 256: delta)
  const float sc = a.scale_log2;
  // dS spill: this tile's unit (scalar base, advanced by 16 KiB per tile: consecutive query tiles of a key block are consecutive units) and
  // the lane's offset inside it
  const unsigned ds_vo = (unsigned)wave * 4096u + (unsigned)lane * 16u;
  const char* ds_base0 = nullptr;
  if constexpr (SPILL)
    ds_base0 = (const char*)a.ds + (((int64_t)b * a.H + h) * ds_units_per_bh(S) + ds_unit(kvblk, 2 * kvblk, S / 64)) * 16384;
  // one (half, k-step) piece: the lane's two 4-query groups -> 8 contiguous queries per lane by a half-wave exchange (on a COPY: the exchange
  // writes both of its operands, and the packed dS is an operand of asm MFMAs hipcc cannot see - exchanged in place right behind the dK chain
  // of the plain tile body, word 0 of a quarter of the lanes came out wrong), then one 16-byte store the compiler can see (it places the wait
  // states an exchange needs in front of a reader itself)
  auto spill_prep = [&](const u32x4_t& d) {
    u32x4_t c = d;
    asm volatile("" : "+v"(c));
    const auto rx = __builtin_amdgcn_permlane32_swap(c[0], c[2], false, false);
    const auto ry = __builtin_amdgcn_permlane32_swap(c[1], c[3], false, false);
    return u32x4_t{rx[0], ry[0], rx[1], ry[1]};
  };
  auto spill_store = [&](const char* unit, auto HF, auto KS, const u32x4_t& o) {
    constexpr int hf = decltype(HF)::value, ks = decltype(KS)::value;
    // (written once, read once by another kernel: non-temporal, so that the stream does not evict the Q / dO tiles the 32 key blocks of a (batch, head) share in L2)
#ifdef MH_SPILL_TEMPORAL
    *(u32x4_t*)(const_cast<char*>(unit) + ds_vo + (hf * 2 + ks) * 1024) = o;
#else
    __builtin_nontemporal_store(o, (u32x4_t*)(const_cast<char*>(unit) + ds_vo + (hf * 2 + ks) * 1024));
#endif
  };
  auto spill = [&](const char* unit, auto HF, auto KS, const u32x4_t& d) {
    if constexpr (SPILL) spill_store(unit, HF, KS, spill_prep(d));
  };
  f32x16_t acck[DBLK], accv[DBLK];
#pragma unroll
  for (int i = 0; i < DBLK; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acck[i][r] = 0.f; accv[i][r] = 0.f; }

  // LDS writes of the tile in the registers into stage NS
  auto wr_q = [&](auto NS_, auto I) {
    constexpr int ns = decltype(NS_)::value, i = decltype(I)::value;
    lds_write128<stage_imm(ns) + 4096 * i>(wb[stage_set(ns)], rq[i]);
  };
  auto wr_do = [&](auto NS_, auto I) {
    constexpr int ns = decltype(NS_)::value, i = decltype(I)::value;
    lds_write128<stage_imm(ns) + OFF_DO + 4096 * i>(wb[stage_set(ns)], rdo[i]);
  };
  auto wr_l = [&](auto NS_) {
    constexpr int ns = decltype(NS_)::value;
    lds_write64<LSE_AREA * ns>(wl, rl);
  };

  // ======== the fast tile body: tile j in stage ST (no masks), tile j + 1 -> stage (ST + 1) % 3, tile j + 2 -> registers ========
  f32x16_t sa[2], pa[2];
  float pv[2][16], dsv[2][16];
  u32x4_t pf[2][2], dsf[2][2];
  auto elem = [&](auto HF, auto R) {
    constexpr int hf = decltype(HF)::value, r = decltype(R)::value;
    pv[hf][r] = fast_exp2(sa[hf][r] * sc);
    dsv[hf][r] = pv[hf][r] * pa[hf][r];
  };
  auto packs = [&](auto HF, auto KS) {
    constexpr int hf = decltype(HF)::value, ks = decltype(KS)::value;
    pf[hf][ks] = pack8v<DT>(pv[hf] + 8 * ks);
    dsf[hf][ks] = pack8v<DT>(dsv[hf] + 8 * ks);
  };
  // ---- ONE continuous fragment stream per tile (64 MFMAs = 64 gaps), rolling across segment and tile seams ---------------------------------
  // MFMA k of a tile: k < 16 S / dP of half 0 (even k: Q fragment x K, odd: dO fragment x V, k-step k / 2), 16 .. 31 the same of half 1,
  // 32 .. 47 dV / dK of half 0 (f = k - 32: tile f % 2 = dO^T -> dV | Q^T -> dK, k-step (f / 2) / 4, d-block (f / 2) % 4), 48 .. 63 of half 1.
  // The fragment of MFMA k is requested at the END of gap k - 6 (row fragments: one ds_read_b128) resp. k - 4 (transposed fragments: two
  // ds_read_b64_tr_b16) - also across the seams: the first transposed fragments in the last gaps of segment B (in front of the barrier: they
  // read tile j, which has been visible since the previous tile's barrier), the next tile's first six row fragments and its half-0 lse / delta
  // reads in the last gaps of segment D (tile j + 1 is visible since this tile's barrier).  No window is ever drained: the four refills per
  // tile, each an exposed LDS round trip, were ~800 of the 3600 cycles of the segment-wise form (profiles/r05_kv_timing.txt).
  // Every s_waitcnt lgkmcnt value comes from Sched::wait(), which walks the instruction order below.
  u32x4_t lsev[2][4], dlv[2][4];
  auto lse_read = [&](auto ST_, auto HF, auto G, auto WHICH) {
    constexpr int st = decltype(ST_)::value, hf = decltype(HF)::value, g = decltype(G)::value, which = decltype(WHICH)::value;
    if constexpr (which == 0) lds_read128<LSE_AREA * st + hf * 128 + 32 * g>(lsev[hf][g], al);
    else lds_read128<LSE_AREA * st + 256 + hf * 128 + 32 * g>(dlv[hf][g], al);
  };
  u32x4_t wr_[6];   // row-fragment window: fragment k (< 32) -> wr_[k % 6]
  u32x2_t wt_[8];   // transposed-fragment window: fragment k (>= 32) -> wt_[2 * (k % 4)], wt_[2 * (k % 4) + 1]
  // request of fragment K (0 .. 63) of the tile in stage ST
  auto frag_req = [&](auto ST_, auto K_) {
    constexpr int st = decltype(ST_)::value, k = decltype(K_)::value;
    if constexpr (k < 32) {
      constexpr int hf = k / 16, n = k % 16;
      constexpr int im = stage_imm(st) + hf * 32 * RB + ((n % 2) ? OFF_DO : 0);
      lds_read128<im>(wr_[k % 6], ar[stage_set(st)][n / 2]);
    } else {
      constexpr int hf = (k - 32) / 16, f = (k - 32) % 16, idx = f / 2, ks = idx / DBLK, db = idx % DBLK;
      constexpr int im = stage_imm(st) + ((f % 2) ? 0 : OFF_DO) + (hf * 32 + ks * 16) * RB;
      lds_read64_tr<im>(wt_[2 * (k % 4)], at[stage_set(st)][2 * db]);
      lds_read64_tr<im + 8 * RB>(wt_[2 * (k % 4) + 1], at[stage_set(st)][2 * db + 1]);
    }
  };
  using H0 = integral_constant<int, 0>;
  using H1 = integral_constant<int, 1>;
  using K0 = integral_constant<int, 0>;
  using K1 = integral_constant<int, 1>;

#ifdef MH_KV_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#endif
  auto tile_fast = [&](auto ST_, int j) {
    constexpr int st = decltype(ST_)::value;
    using NS = integral_constant<int, (st + 1) % 3>;
    using SCH = kv3::Sched;
    TileSrd ts;
    static_for<64>([&](auto G_) {
      constexpr int g = decltype(G_)::value;
      lgkm_wait<SCH::wait(g)>();
      if constexpr (g == 0 || g == 16) {  // the half's -lse / scale and -delta are the accumulators' initial values
        constexpr int hf = g / 16;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            sa[hf][4 * q + e] = __uint_as_float(lsev[hf][q][e]);
            pa[hf][4 * q + e] = __uint_as_float(dlv[hf][q][e]);
          }
        mfma_ready(sa[hf], pa[hf]);
      }
      if constexpr (g < 32) {
        constexpr int hf = g / 16, n = g % 16;
        if constexpr (n % 2 == 0) mfma32va<DT>(sa[hf], wr_[g % 6], kf[n / 2]);
        else mfma32va<DT>(pa[hf], wr_[g % 6], vf[n / 2]);
      } else {
        constexpr int hf = (g - 32) / 16, f = (g - 32) % 16, idx = f / 2, ks = idx / DBLK, db = idx % DBLK;
        const u32x4_t fr = u32x4_t{wt_[2 * (g % 4)][0], wt_[2 * (g % 4)][1], wt_[2 * (g % 4) + 1][0], wt_[2 * (g % 4) + 1][1]};
        if constexpr (f % 2 == 0) mfma32a<DT>(accv[db], fr, pf[hf][ks]);
        else mfma32a<DT>(acck[db], fr, dsf[hf][ks]);
      }
      __builtin_amdgcn_sched_barrier(0);  // (the gap's fillers go BEHIND its MFMA: hipcc is free to order plain VALU around an asm MFMA it cannot see)
      // ---- fillers, VALU / VMEM first ----
      // the registers (tile j + 1) have been in flight for a whole tile (the previous tile's four dS stores are younger: they may stay in flight)
      if constexpr (g == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SPILL ? 4 : 0) : "memory");
      if constexpr (g == 3) ts = tile_srd(j + 2);  // (scalar arithmetic behind an MFMA, well ahead of the loads that read the descriptors)
      if constexpr (g >= 9 && g < 13) ld_q(integral_constant<int, g - 9>{}, ts);        // requests of tile j + 2 (the registers are free from gap 9 on)
      if constexpr (g >= 13 && g < 17) ld_do(integral_constant<int, g - 13>{}, ts);
      if constexpr (g == 17) ld_l(ts);
      // exp / dS of half 0: 16 elements over gaps 18 .. 31 (its last MFMAs are gaps 14 / 15: the MFMA -> VALU wait states are covered)
      if constexpr (g >= 18 && g < 32) {
        constexpr int r0 = ((g - 18) * 16) / 14, r1 = ((g - 17) * 16) / 14;
        static_for<r1 - r0>([&](auto R) { elem(H0{}, integral_constant<int, r0 + decltype(R)::value>{}); });
      }
      if constexpr (g == 28) packs(H0{}, K0{});
      if constexpr (g == 32) packs(H0{}, K1{});
      // half 1: elements 0 .. 9 over gaps 34 .. 43, 10 .. 15 behind the k-step-0 MFMAs of its own dV / dK segment (gaps 48 .. 53)
      if constexpr (g >= 34 && g < 44) elem(H1{}, integral_constant<int, g - 34>{});
      if constexpr (g == 45) packs(H1{}, K0{});
      if constexpr (g >= 48 && g < 54) elem(H1{}, integral_constant<int, g - 38>{});
      if constexpr (g == 54) packs(H1{}, K1{});
      if constexpr (SPILL) {  // dS of this tile -> its 16-KiB unit (gaps without element work)
        const char* unit = ds_base0 + (int64_t)j * 16384;
        if constexpr (g == 44) spill(unit, H0{}, K0{}, dsf[0][0]);
        if constexpr (g == 46) spill(unit, H0{}, K1{}, dsf[0][1]);
        if constexpr (g == 47) spill(unit, H1{}, K0{}, dsf[1][0]);
        if constexpr (g == 56) spill(unit, H1{}, K1{}, dsf[1][1]);
      }
      // ---- LDS fillers (Sched::fillers) ----
      if constexpr (g < 4) wr_q(NS{}, integral_constant<int, g>{});
      else if constexpr (g < 8) wr_do(NS{}, integral_constant<int, g - 4>{});
      else if constexpr (g == 8) wr_l(NS{});
      if constexpr (g >= 9 && g < 13) {  // half 1's lse / delta reads (its accumulator registers are dead since the previous tile's segment D)
        lse_read(ST_, H1{}, integral_constant<int, (2 * (g - 9)) % 4>{}, integral_constant<int, (2 * (g - 9)) / 4>{});
        lse_read(ST_, H1{}, integral_constant<int, (2 * (g - 9) + 1) % 4>{}, integral_constant<int, (2 * (g - 9) + 1) / 4>{});
      }
      if constexpr (g >= 50 && g < 58)  // the NEXT tile's half-0 lse / delta reads
        lse_read(NS{}, H0{}, integral_constant<int, (g - 50) % 4>{}, integral_constant<int, (g - 50) / 4>{});
      // ---- fragment requests: transposed fragment g + 4, row fragment g + 6 (of the next tile from gap 58 on) ----
      if constexpr (g + 4 >= 32 && g + 4 < 64) frag_req(ST_, integral_constant<int, g + 4>{});
      if constexpr (g + 6 < 32) frag_req(ST_, integral_constant<int, g + 6>{});
      if constexpr (g + 6 >= 64) frag_req(NS{}, integral_constant<int, g + 6 - 64>{});
      if constexpr (g == 31) {  // every LDS write of tile j + 1 is older than a read this wave has already waited for
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    });
#ifdef MH_KV_TIMING
    tacc[6] += 1;
    KV_STAMP(0);
#endif
  };
  // what the fast body expects in flight when it is entered: the LDS instruction sequence of gaps 50 .. 63 of a preceding fast tile (the
  // transposed reads of that tile are stand-ins here - same count, same order, results unused - so that the counted waits hold)
  auto fast_prologue = [&](auto ST_) {
    static_for<4>([&](auto G) { lse_read(ST_, H0{}, G, integral_constant<int, 0>{}); });
    static_for<4>([&](auto G) { lse_read(ST_, H0{}, G, integral_constant<int, 1>{}); });
    frag_req(ST_, integral_constant<int, 62>{});
    frag_req(ST_, integral_constant<int, 0>{});
    frag_req(ST_, integral_constant<int, 63>{});
    frag_req(ST_, integral_constant<int, 1>{});
    static_for<4>([&](auto I) { frag_req(ST_, integral_constant<int, 2 + decltype(I)::value>{}); });
  };

  // ======== the plain tile body (run-time stage; masks where the tile needs them): the registers -> stage ns and the requests of tile j + 2
  // first, then the barrier, then the two 32-query halves one after the other (S, dP, exp / dS, dV, dK) ========
  auto write_regs = [&](int ns) {  // all nine LDS writes of the tile in the registers into stage ns
    const unsigned tb = lds0 + TILES0 + (unsigned)ns * STG + (unsigned)tid * 16u, lb = wl + (unsigned)ns * LSE_AREA;
    static_for<4>([&](auto I) {
      constexpr int i = decltype(I)::value;
      lds_write128<4096 * i>(tb, rq[i]);
      lds_write128<OFF_DO + 4096 * i>(tb, rdo[i]);
    });
    lds_write64<0>(lb, rl);
  };
  auto request = [&](int j) {  // all nine loads of tile j into the registers
    const TileSrd ts = tile_srd(j);
    asm volatile("s_nop 4" ::: "memory");  // (descriptor words may have come through v_readfirstlane: cdna guide, inline asm item 2)
    static_for<4>([&](auto I) { ld_q(I, ts); ld_do(I, ts); });
    ld_l(ts);
  };
  auto tile_gen = [&](int j, int st) {
    const int q0 = q_begin + j * 64;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    write_regs(st == 2 ? 0 : st + 1);
    request(j + 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    unsigned aq[KSTEPS], ado[KSTEPS], atq[KSTEPS], atdo[KSTEPS];
    const unsigned so = (unsigned)st * STG;
#pragma unroll
    for (int i = 0; i < KSTEPS; ++i) {
      aq[i] = ar[0][i] + so;
      ado[i] = aq[i] + OFF_DO;
      atq[i] = at[0][i] + so;
      atdo[i] = atq[i] + OFF_DO;
    }
    const unsigned alr = al + (unsigned)st * LSE_AREA;
    static_for<2>([&](auto HALF) {
      constexpr int hf = decltype(HALF)::value;
      const int qh = q0 + 32 * hf;
      const char* unit = SPILL ? ds_base0 + (int64_t)j * 16384 : nullptr;
      if ((CAUSAL && kvw0 > qh + 31) || qh >= len) {  // this wave's keys see none of these queries (wave-uniform)
        const u32x4_t z = {0u, 0u, 0u, 0u};
        spill(unit, HALF, K0{}, z);
        spill(unit, HALF, K1{}, z);
        return;
      }
      f32x16_t sacc, pacc;
      u32x4_t lv[4], dv4[4];
      static_for<4>([&](auto I) { constexpr int g = decltype(I)::value; lds_read128<hf * 128 + 32 * g>(lv[g], alr); });
      stream_row_frags<hf * 32 * RB, KSTEPS>(aq, [&](auto I, const u32x4_t& fr) {
        if constexpr (decltype(I)::value == 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) sacc[4 * g + e] = __uint_as_float(lv[g][e]);
          mfma_ready(sacc, sacc);
        }
        mfma32va<DT>(sacc, fr, kf[decltype(I)::value]);
      });
      static_for<4>([&](auto I) { constexpr int g = decltype(I)::value; lds_read128<256 + hf * 128 + 32 * g>(dv4[g], alr); });
      stream_row_frags<hf * 32 * RB, KSTEPS, 4>(ado, [&](auto I, const u32x4_t& fr) {
        if constexpr (decltype(I)::value == 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) pacc[4 * g + e] = __uint_as_float(dv4[g][e]);
          mfma_ready(pacc, pacc);
        }
        mfma32va<DT>(pacc, fr, vf[decltype(I)::value]);
      });
      mfma_settle(sacc, pacc);
      float p_[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) p_[r] = fast_exp2(sacc[r] * sc);
      if ((qh + 32 > len) || (kvw0 + 32 > len) || (CAUSAL && (kvw0 + 31 > qh))) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int q = qh + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const bool ok = (q < len) && (kvrow < len) && (!CAUSAL || kvrow <= q);
          p_[r] = ok ? p_[r] : 0.f;
        }
      }
      {
        const u32x4_t pfr[2] = {pack8v<DT>(p_), pack8v<DT>(p_ + 8)};
        stream_tr_frags<RB, hf * 32, 2 * DBLK>(atdo, [&](auto I, const u32x4_t& fr) {
          constexpr int f = decltype(I)::value;
          mfma32a<DT>(accv[f / 2], fr, pfr[f % 2]);
        });
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) p_[r] *= pacc[r];  // dS (unscaled)
      {
        const u32x4_t dsr[2] = {pack8v<DT>(p_), pack8v<DT>(p_ + 8)};
        u32x4_t so0, so1;
        if constexpr (SPILL) { so0 = spill_prep(dsr[0]); so1 = spill_prep(dsr[1]); }
        stream_tr_frags<RB, hf * 32, 2 * DBLK>(atq, [&](auto I, const u32x4_t& fr) {
          constexpr int f = decltype(I)::value;
          mfma32a<DT>(acck[f / 2], fr, dsr[f % 2]);
        });
        if constexpr (SPILL) { spill_store(unit, HALF, K0{}, so0); spill_store(unit, HALF, K1{}, so1); }
      }
    });
  };

  // ---- tiles: diagonal tiles (masks), then the fully visible ones, then a tail at the sequence end.  Tile j sits in stage (j + soff) % 3,
  // soff chosen so that the first fully visible tile is in stage 0 (the fast body's stage is a template parameter: unrolled by 3) ----
  const bool keys_full = (kv0 + 128 <= len);
  const int n_diag = CAUSAL ? min(ntiles, 2) : 0;
  const int n_tail = (len % 64) ? 1 : 0;
  const int j_full_end = keys_full ? max(n_diag, ntiles - n_tail) : n_diag;
  const int soff = (3 - n_diag % 3) % 3;
  auto stage_of = [&](int j) { return (j + soff) % 3; };
  request(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  write_regs(stage_of(0));
  request(1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  int j = 0;
  for (; j < n_diag; ++j) tile_gen(j, stage_of(j));
#ifdef MH_KV_TIMING
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev) :: "memory");
#endif
  if (j < j_full_end) {
    fast_prologue(integral_constant<int, 0>{});
    for (;;) {  // (three exits, no merge back into the loop: every fully visible tile takes the fast body)
      tile_fast(integral_constant<int, 0>{}, j);
      if (++j >= j_full_end) break;
      tile_fast(integral_constant<int, 1>{}, j);
      if (++j >= j_full_end) break;
      tile_fast(integral_constant<int, 2>{}, j);
      if (++j >= j_full_end) break;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the last tile's run-ahead requests for a tile the fast body does not run)
    __builtin_amdgcn_sched_barrier(0);
  }
#ifdef MH_KV_TIMING
  KV_STAMP(7);
  if (a.dbg && lane == 0)
    for (int i = 0; i < 8; ++i) a.dbg[((int64_t)blockIdx.x * 4 + wave) * 8 + i] = tacc[i];
#endif
  for (; j < ntiles; ++j) tile_gen(j, stage_of(j));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the last requests - tiles past the end, zeros - are not left in flight)

#pragma unroll
  for (int i = 0; i < DBLK; ++i) { mfma_settle_acc(accv[i]); mfma_settle_acc(acck[i]); }
  if (kvrow < S) {
    const bool valid = kvrow < len;
    auto store_rows = [&](auto& acc, uint16_t* outp) {
      auto val = [&](int i, int r) { return acc[i][r]; };
      if (a.wide) store_row_wide<DT, DBLK>(outp, hi, valid, val);
      else store_row_narrow<DT, DBLK>(outp, hi, valid, val);
    };
    store_rows(accv, outv);
#pragma unroll
    for (int i = 0; i < DBLK; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acck[i][r] *= a.scale;
    if (a.rope && valid) unrope_rows<D>(acck, a.rope + (int64_t)kvrow * (D / 2), hi);
    store_rows(acck, outk);
  }
}


// ------------------------------------------------------------------------------------------------------------
// dQ from the spilled dS (5-product backward): dQ^T[d, q] = sum_k K^T[d, k] dS^T[k, q], one product, no S / dP recomputation, no exp
// ------------------------------------------------------------------------------------------------------------
// One block per 128 queries (4 waves x 32 query columns), key tiles of 64: per tile 16 MFMAs per wave against 16 KiB of K (L2-resident: shared by
// the 32 query blocks of a (batch, head)) and 16 KiB of dS that is read from HBM exactly once over the launch - a 4.4-GB read stream at cfg 3.  A
// tile is ~0.3 us of matrix work, far below any memory latency, so everything is requested THREE tiles ahead and nothing is waited for at a seam:
//   * K tiles by LDS-DMA into a ring of three stages, two tiles ahead (one barrier per tile orders the ring);
//   * dS through registers, three tiles ahead (three register sets; the tile loop is unrolled by three).  A wave needs exactly the dS of ITS 32
//     query columns: four 1-KiB pieces of the writer's layout (ds_unit), fetched with coalesced 16-byte loads and written into the wave's own
//     columns of a row-major [64 keys][128 queries] tile (same swizzle as every tile here) - wave-private data: the LDS executes one wave's
//     instructions in order, so no barrier orders these writes and reads.  The B operand then comes from the same transpose-reads that give K^T
//     from the K tile: both operands share the slot <-> key order by construction.
// Requests past the last tile re-fetch the last tile (harmless; keeps every s_waitcnt count a constant).  Scale + inverse RoPE in the epilogue.
template <int DT>
__global__ __launch_bounds__(256, 2) void attn_bwd3_dq_k(Bwd2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int D = 128, RB = 256, T_BYTES = 64 * RB, DBLK = 4;
  constexpr int OFF_DS = 3 * T_BYTES;  // [K stage 0 | K stage 1 | K stage 2 | dS tile]
  using std::integral_constant;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int S = a.S, nq = S / 128, nqt = S / 64;
  int bh_, qi;
  if (!xcd_work(a.B * a.H, nq, bh_, qi)) return;
  const int qblk = nq - 1 - qi;  // longest blocks first
  const int h = bh_ % a.H, b = bh_ / a.H;
  const int q0 = qblk * 128, qrow = q0 + wave * 32 + l31;
  uint16_t* dqp = a.dq + ((int64_t)b * S + qrow) * a.lddq + (int64_t)h * D;
  const int ntiles = 2 * qblk + 2;  // key tiles 0 .. (q0 + 127) / 64
  const uint16_t* kbase = a.k + (int64_t)b * S * a.ldk + (int64_t)h * D;
  const unsigned lds0 = lds_addr_of(smem);
  const auto src_k = row_src<D>(kbase, a.ldk, S, tid);
  auto stage_k = [&](int st, int kt) { stage_rows_buf<D, 64>(src_k, min(kt, ntiles - 1) * 64, lds0 + (unsigned)st * T_BYTES + (unsigned)wave * 1024u); };
  // this wave's dS pieces of key tile kt: query tile 2 qblk + (wave >> 1); piece pi -> writer wave 2 (kt & 1) + (pi >> 1), (half, k-step) 2 (wave & 1) + (pi & 1)
  const int qtsel = wave >> 1, u0 = 2 * (wave & 1);
  const char* ds_bh = (const char*)a.ds + ((int64_t)b * a.H + h) * ds_units_per_bh(S) * 16384;
  auto piece0 = [&](int kt) {
    kt = min(kt, ntiles - 1);
    return ds_bh + (int64_t)ds_unit(kt >> 1, 2 * qblk + qtsel, nqt) * 16384 + (2 * (kt & 1)) * 4096 + u0 * 1024 + lane * 16;
  };
  // LDS positions of the lane's four chunks: row 32 (pi >> 1) + l31, logical chunk c = 8 qtsel + 2 (u0 + (pi & 1)) + hi
  char* wds[4];
#pragma unroll
  for (int pi = 0; pi < 4; ++pi) {
    const int row = 32 * (pi >> 1) + l31, c = 8 * qtsel + 2 * (u0 + (pi & 1)) + hi;
    wds[pi] = smem + OFF_DS + row * RB + ((c ^ TileSwz<D>::f(row)) << 4);
  }
  unsigned off_t[D / 16];
  tr_frag_offsets<D>(lane, off_t);
  const unsigned ad0 = lds0 + OFF_DS + tr_frag_offset_one<D>(lane, wave, 0), ad1 = lds0 + OFF_DS + tr_frag_offset_one<D>(lane, wave, 1);
  f32x16_t dqacc[DBLK];
#pragma unroll
  for (int i = 0; i < DBLK; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqacc[i][r] = 0.f;
  u32x4_t rs[3][4];  // dS of tiles j, j + 1, j + 2 (set = tile % 3)
  auto load = [&](auto SET, int kt) {
    constexpr int st = decltype(SET)::value;
    const char* pp = piece0(kt);
    const char* pc = pp + 2048;            // (the immediate offset of a global load is 13 bits, signed)
    gload128<-2048>(rs[st][0], pc);
    gload128<-1024>(rs[st][1], pc);
    gload128<2048>(rs[st][2], pc);         // + 4096: the writer's next wave = the other 32 keys of the tile
    gload128<3072>(rs[st][3], pc);
  };
  auto tile = [&](int j, auto ST_) {  // ST = j % 3: K stage and dS register set of tile j
    constexpr int st = decltype(ST_)::value;
    // K of tile j (requested two tiles ago) has landed when at most the 12 younger requests are in flight (dS j + 1, K j + 1, dS j + 2)
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    stage_k((st + 2) % 3, j + 2);  // (that stage held tile j - 1: every wave has left it)
    // this wave's dS columns of tile j: registers -> LDS (the set was requested three tiles ago: older than K of tile j), then the set is refilled with tile j + 3
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) *(u32x4_t*)wds[pi] = rs[st][pi];
    load(ST_, j + 3);
    const unsigned sb = lds0 + (unsigned)st * T_BYTES;
    unsigned ak[D / 16];
#pragma unroll
    for (int i = 0; i < D / 16; ++i) ak[i] = sb + off_t[i];
    // B fragments: dS^T, query block = this wave, k-steps 0 .. 3 over the 64 keys
    u32x2_t bw[8];
    static_for<4>([&](auto I) {
      constexpr int ks = decltype(I)::value;
      lds_read64_tr<(ks * 16) * RB>(bw[2 * ks], ad0);
      lds_read64_tr<(ks * 16 + 8) * RB>(bw[2 * ks + 1], ad1);
    });
    // A fragments: K^T, f = (d-block f / 4, k-step f % 4), rolling window of 3 (8 + 6 reads in flight: the lgkm counter holds 15)
    u32x2_t w[6];
    auto issue = [&](auto F, auto SLOT) {
      constexpr int f = decltype(F)::value, sl = decltype(SLOT)::value, i = f / 4, ks = f % 4;
      lds_read64_tr<(ks * 16) * RB>(w[2 * sl], ak[2 * i]);
      lds_read64_tr<(ks * 16 + 8) * RB>(w[2 * sl + 1], ak[2 * i + 1]);
    };
    static_for<3>([&](auto I) { issue(I, I); });
    static_for<16>([&](auto I) {
      constexpr int f = decltype(I)::value, i = f / 4, ks = f % 4;
      constexpr int left = 15 - f;
      lgkm_wait<2 * (left < 2 ? left : 2)>();
      const u32x4_t fa = u32x4_t{w[2 * (f % 3)][0], w[2 * (f % 3)][1], w[2 * (f % 3) + 1][0], w[2 * (f % 3) + 1][1]};
      const u32x4_t fb = u32x4_t{bw[2 * ks][0], bw[2 * ks][1], bw[2 * ks + 1][0], bw[2 * ks + 1][1]};
      dqacc[i] = mfma32v<DT>(fa, fb, dqacc[i]);
      if constexpr (f + 3 < 16) issue(integral_constant<int, f + 3>{}, integral_constant<int, f % 3>{});
    });
  };
  // prologue: the request order of three tile tops (... dS 0 | K 0, dS 1 | K 1, dS 2), so that the counted wait of tile 0 holds
  load(integral_constant<int, 0>{}, 0);
  stage_k(0, 0);
  load(integral_constant<int, 1>{}, 1);
  stage_k(1, 1);
  load(integral_constant<int, 2>{}, 2);
  for (int j = 0;;) {
    tile(j, integral_constant<int, 0>{});
    if (++j >= ntiles) break;
    tile(j, integral_constant<int, 1>{});
    if (++j >= ntiles) break;
    tile(j, integral_constant<int, 2>{});
    if (++j >= ntiles) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (run-ahead requests of tiles that do not exist)
  {
#pragma unroll
    for (int i = 0; i < DBLK; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) dqacc[i][r] *= a.scale;
    if (a.rope) unrope_rows<D>(dqacc, a.rope + (int64_t)qrow * (D / 2), hi);
    auto val = [&](int i, int r) { return dqacc[i][r]; };
    if (a.wide) store_row_wide<DT, DBLK>(dqp, hi, true, val);
    else store_row_narrow<DT, DBLK>(dqp, hi, true, val);
  }
}

// dK + dV in one kernel (S and dP computed once per tile: 4 instead of 5 products for the pair); A-B switch mh_attn_bwd_fused_kv
int g_attn_bwd_fused_kv = 2;  // 0: two kernels, 1: attn_bwd2_kv_k<MODE 3> (rounds 2-4), 2: attn_bwd3_kv_k (default)

template <int DT, int D, bool CAUSAL>
int launch_bwd2(const Bwd2Args& a, hipStream_t st) {
  constexpr size_t ldsQ = 2 * 2 * 64 * D * 2, ldsKV = 2 * (2 * 64 * D * 2 + 2048);
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)attn_bwd2_dq_k<DT, D, CAUSAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsQ);
    hipFuncSetAttribute((const void*)attn_bwd2_kv_k<DT, D, CAUSAL, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsKV);
    hipFuncSetAttribute((const void*)attn_bwd2_kv_k<DT, D, CAUSAL, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsKV);
    hipFuncSetAttribute((const void*)attn_bwd2_kv_k<DT, D, CAUSAL, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsKV);
    attr = true;
  }
  const int64_t nth = (int64_t)a.B * a.S * a.H;
  constexpr int RPB = 256 / (D / 8);
  hipLaunchKernelGGL((delta2_k<DT, D>), dim3((unsigned)((nth + RPB - 1) / RPB)), dim3(256), 0, st, a);
  dim3 grid(xcd_grid(a.B * a.H, (a.S + 127) / 128));
  bool kv3_done = false;
  if constexpr (D == 128 && CAUSAL) {  // (the non-causal instantiation of attn_bwd3_kv_k spills registers: D = 128 without a mask - not a shape of
                                       //  this model - stays on attn_bwd2_kv_k<MODE 3>)
    if (g_attn_bwd_fused_kv == 2) {  // round-5 form: register-staged copies, three LDS stages, one barrier in the middle of a tile
      static bool attr3 = false;
      if (!attr3) {
        hipFuncSetAttribute((const void*)attn_bwd3_kv_k<DT, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kv3::LDS_BYTES);
        hipFuncSetAttribute((const void*)attn_bwd3_kv_k<DT, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kv3::LDS_BYTES);
        hipFuncSetAttribute((const void*)attn_bwd3_dq_k<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsQ);
        attr3 = true;
      }
      if (a.ds && a.S % 128 == 0 && !a.seqlens) {  // 5-product form: dS spilled by the dK|dV kernel, dQ = dS K as a one-product pass
        hipLaunchKernelGGL((attn_bwd3_kv_k<DT, true, true>), grid, dim3(256), kv3::LDS_BYTES, st, a);
        hipLaunchKernelGGL((attn_bwd3_dq_k<DT>), grid, dim3(256), ldsQ, st, a);
        MH_LAUNCH_CHECK();
      }
      hipLaunchKernelGGL((attn_bwd3_kv_k<DT, true, false>), grid, dim3(256), kv3::LDS_BYTES, st, a);
      kv3_done = true;
    }
  }
  if (kv3_done) {
  } else if (g_attn_bwd_fused_kv && D == 128) {  // (D = 64, the vision tower's 577-token sequences: measured slower fused, 0.48 vs 0.40 ms)
    hipLaunchKernelGGL((attn_bwd2_kv_k<DT, D, CAUSAL, 3>), grid, dim3(256), ldsKV, st, a);
  } else {
    hipLaunchKernelGGL((attn_bwd2_kv_k<DT, D, CAUSAL, 1>), grid, dim3(256), ldsKV, st, a);
    hipLaunchKernelGGL((attn_bwd2_kv_k<DT, D, CAUSAL, 2>), grid, dim3(256), ldsKV, st, a);
  }
  hipLaunchKernelGGL((attn_bwd2_dq_k<DT, D, CAUSAL>), grid, dim3(256), ldsQ, st, a);
  MH_LAUNCH_CHECK();
}

}  // namespace
}  // namespace mhattn

extern "C" void mh_attn_bwd_fused_kv(int on) { mhattn::g_attn_bwd_fused_kv = on < 0 ? 0 : (on > 2 ? 2 : on); }
extern "C" int mh_attn_bwd_fused_kv_mode(void) { return mhattn::g_attn_bwd_fused_kv; }
#ifdef MH_KV_TIMING
static unsigned long long* g_kv_timing_dbg = nullptr;
extern "C" void mh_kv_timing_buffer(void* p) { g_kv_timing_dbg = (unsigned long long*)p; }
#endif

static int attn_bwd_impl(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o,
                         int64_t ldo, const void* dout, int64_t lddo, const float* lse, float* delta, void* dq, int64_t lddq,
                         void* dk, int64_t lddk, void* dv, int64_t lddv, const int32_t* seqlens, int B, int S, int H, int D,
                         int causal, const float* rope_cos_sin, int dt, void* ds_ws, void* stream) {
  using namespace mhattn;
  if (!q || !k || !v || !o || !dout || !lse || !delta || !dq || !dk || !dv) return MH_ERR_ARG;
  if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (lddo & 7) || (ldo & 7) || (lddq & 3) || (lddk & 3) || (lddv & 3)) return MH_ERR_ARG;
  for (int64_t ld_ : {ldq, ldk, ldv, lddo})
    if ((int64_t)S * ld_ * 2 >= (1ll << 31)) return MH_ERR_SHAPE;  // one batch element under a 31-bit num_records (stage_rows_buf)
  if (D != 128 && D != 64) return MH_ERR_SHAPE;
  Bwd2Args a;
  a.q = (const uint16_t*)q; a.k = (const uint16_t*)k; a.v = (const uint16_t*)v; a.o = (const uint16_t*)o;
  a.dout = (const uint16_t*)dout; a.lse = lse; a.delta = delta;
  a.B = B; a.S = S; a.H = H; a.S_pad = (S + 63) / 64 * 64;
  a.lse2 = delta + (int64_t)B * H * a.S_pad;
  a.dq = (uint16_t*)dq; a.dk = (uint16_t*)dk; a.dv = (uint16_t*)dv; a.seqlens = seqlens;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddo = lddo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  a.scale = 1.0f / sqrtf((float)D);
  a.scale_log2 = a.scale * 1.4426950408889634f;
  a.inv_scale = sqrtf((float)D);
  a.rope = (const float2*)rope_cos_sin;
  a.ds = (uint16_t*)ds_ws;
  a.wide = ((lddq & 7) == 0) && ((lddk & 7) == 0) && ((lddv & 7) == 0) && aligned16(dq) && aligned16(dk) && aligned16(dv) && g_attn_wide_stores;
#ifdef MH_KV_TIMING
  a.dbg = g_kv_timing_dbg;
#endif
  hipStream_t st = as_stream(stream);
#define GO(DT_, D_, C_) return launch_bwd2<DT_, D_, C_>(a, st)
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

extern "C" int mh_attn_bwd2(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o,
                            int64_t ldo, const void* dout, int64_t lddo, const float* lse, float* delta, void* dq, int64_t lddq,
                            void* dk, int64_t lddk, void* dv, int64_t lddv, const int32_t* seqlens, int B, int S, int H, int D,
                            int causal, const float* rope_cos_sin, int dt, void* stream) {
  return attn_bwd_impl(q, ldq, k, ldk, v, ldv, o, ldo, dout, lddo, lse, delta, dq, lddq, dk, lddk, dv, lddv, seqlens, B, S, H, D, causal, rope_cos_sin,
                       dt, nullptr, stream);
}
extern "C" int64_t mh_attn_bwd_spill_bytes(int B, int S, int H) { return (int64_t)B * H * mhattn::ds_units_per_bh(S) * 16384; }
extern "C" int mh_attn_bwd2_spill(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o,
                                  int64_t ldo, const void* dout, int64_t lddo, const float* lse, float* delta, void* dq, int64_t lddq,
                                  void* dk, int64_t lddk, void* dv, int64_t lddv, const int32_t* seqlens, int B, int S, int H, int D,
                                  int causal, const float* rope_cos_sin, int dt, void* ds_ws, void* stream) {
  return attn_bwd_impl(q, ldq, k, ldk, v, ldv, o, ldo, dout, lddo, lse, delta, dq, lddq, dk, lddk, dv, lddv, seqlens, B, S, H, D, causal, rope_cos_sin,
                       dt, ds_ws, stream);
}
