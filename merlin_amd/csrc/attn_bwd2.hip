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

// dK + dV in one kernel (S and dP computed once per tile: 4 instead of 5 products for the pair); A-B switch mh_attn_bwd_fused_kv
int g_attn_bwd_fused_kv = 1;

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
  if (g_attn_bwd_fused_kv && D == 128) {  // (D = 64, the vision tower's 577-token sequences: measured slower fused, 0.48 vs 0.40 ms)
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

extern "C" void mh_attn_bwd_fused_kv(int on) { mhattn::g_attn_bwd_fused_kv = on ? 1 : 0; }
#ifdef MH_KV_TIMING
static unsigned long long* g_kv_timing_dbg = nullptr;
extern "C" void mh_kv_timing_buffer(void* p) { g_kv_timing_dbg = (unsigned long long*)p; }
#endif

extern "C" int mh_attn_bwd2(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o,
                            int64_t ldo, const void* dout, int64_t lddo, const float* lse, float* delta, void* dq, int64_t lddq,
                            void* dk, int64_t lddk, void* dv, int64_t lddv, const int32_t* seqlens, int B, int S, int H, int D,
                            int causal, const float* rope_cos_sin, int dt, void* stream) {
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
