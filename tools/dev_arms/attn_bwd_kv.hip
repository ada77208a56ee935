// Attention backward, KV-block-outer kernels: dV and dK (gfx950).  Design notes: attn_bwd.hip.
//
// One block per 128 keys of one (b,h) (4 waves x 32 keys; K - and for dK also V - fragments live in
// registers as MFMA B operands), streaming 32-query tiles through a double-buffered LDS stage filled by
// global_load_lds.  MODE selects what a launch accumulates:
//     MODE 1 (dV):  S = Q K^T -> P;                     dV^T += dO^T P      (16 MFMA 32x32x16 / tile / wave)
//     MODE 2 (dK):  S, dP = dO V^T -> dS = P o (dP - delta) * scale;  dK^T += Q^T dS   (24 MFMA)
//     MODE 3 (both in one pass, 32 MFMA, but ~270 live VGPRs -> spills; kept for A/B measurements)
// Splitting into MODE 1 + MODE 2 recomputes S once (+25 % MFMAs) and buys two lean kernels (<= 192 VGPRs,
// no spills, >= 2 waves per SIMD) - the better trade on this chip.
//
// The A-operand fragments (rows of Q / dO / Q^T / dO^T tiles, and the lse / delta vectors) are read from LDS
// with hand-waited inline-asm ds_read_b128 (mh_common.h), so the next tile's global_load_lds stay in flight
// under the whole compute of the current tile; the only vmcnt(0) is the one in front of the per-tile barrier.
#include <type_traits>

#include "attn_bwd_common.h"

namespace mhattn {
namespace {

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

template <int DT, int D, bool CAUSAL, int MODE>
__global__ __launch_bounds__(256, 2) void attn_bwd_kv_k(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool DO_DV = (MODE & 1) != 0, DO_DK = (MODE & 2) != 0;
  constexpr int CPR = D / 8;               // 16-byte chunks per row-major row
  constexpr int TILE = 32 * D * 2;         // bytes of one [32][D] (or [D][32]) tile
  constexpr int OFF_Q = 0, OFF_DO = TILE, OFF_QT = 2 * TILE, OFF_DOT = (DO_DK ? 3 : 1) * TILE;
  // ring of NS stages; a stage holds only the tiles this MODE reads (+ 4 per-wave copies of lse|delta, so every
  // thread issues the same number of loads per tile and the pipeline wait can be a COUNTED vmcnt)
  constexpr int NKIND = 1 + (DO_DK ? 2 : 0) + (DO_DV ? 1 : 0);
  constexpr int STAGE = NKIND * TILE + 1024;
  constexpr int NS = (MODE == 1) ? 4 : 3;
  constexpr int KSTEPS = D / 16, DBLK = D / 32;
  constexpr int NLD = TILE / (256 * 16);   // glds per thread per tile kind
  constexpr int LOADS_PER_TILE = NKIND * NLD + 1;
  constexpr int NFR = (KSTEPS > 2 * DBLK) ? KSTEPS : 2 * DBLK;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  int bh_, kvblk;
  if (!xcd_work(a.B * a.H, (a.S + 127) / 128, bh_, kvblk)) return;  // kv-block 0 (most query tiles) first
  const int h = bh_ % a.H, b = bh_ / a.H;
  const int S = a.S;
  const int len = a.seqlens ? min(a.seqlens[b], S) : S;
  const int kv0 = kvblk * 128;
  const int kvrow = kv0 + wave * 32 + l31;
  uint16_t* dkp = a.dk + ((int64_t)b * S + kvrow) * a.lddk + (int64_t)h * D;
  uint16_t* dvp = a.dv + ((int64_t)b * S + kvrow) * a.lddv + (int64_t)h * D;

  if (kv0 >= len) {
    if (kvrow < S) {
      for (int d = hi * (D / 2); d < (hi + 1) * (D / 2); d += 4) {
        if (DO_DK) *(uint2*)(dkp + d) = make_uint2(0, 0);
        if (DO_DV) *(uint2*)(dvp + d) = make_uint2(0, 0);
      }
    }
    return;
  }

  // K, V fragments (B operands): lane holds X[kvrow][16*ks + 8*hi .. +8]
  u32x4_t kf[KSTEPS], vf[DO_DK ? KSTEPS : 1];
  {
    const int kr = min(kvrow, S - 1);
    const uint16_t* kp = a.k + ((int64_t)b * S + kr) * a.ldk + (int64_t)h * D + 8 * hi;
    const uint16_t* vp = a.v + ((int64_t)b * S + kr) * a.ldv + (int64_t)h * D + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      kf[ks] = *(const u32x4_t*)(kp + 16 * ks);
      if constexpr (DO_DK) vf[ks] = *(const u32x4_t*)(vp + 16 * ks);
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
  const float* lse_row = a.lse2 + ((int64_t)b * a.H + h) * a.S_pad;  // exp2-domain lse
  const float* dl_row = a.delta + ((int64_t)b * a.H + h) * a.S_pad;
  constexpr int OFF_LSE = NKIND * TILE;  // + wave * 256: [lse 32 f32 | delta 32 f32]
  auto stage = [&](int jt) {  // tile index jt (clamped: the tail re-loads the last tile into a dead slot)
    const int q0 = q_begin + min(jt, ntiles - 1) * 32;
    char* base = smem + (jt % NS) * STAGE;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int qr = min(q0 + rrow[i], S - 1);
      const int64_t roff = ((int64_t)b * S + qr);
      char* dst = base + (i * 256 + wave * 64) * 16;
      glds16(a.q + roff * a.ldq + (int64_t)h * D + rcol[i], dst + OFF_Q);
      if (DO_DK) glds16(a.dout + roff * a.lddo + (int64_t)h * D + rcol[i], dst + OFF_DO);
      if (DO_DK) glds16(a.qt + (bh_t + trow[i]) * a.S_pad + q0 + tcol[i], dst + OFF_QT);
      if (DO_DV) glds16(a.dot + (bh_t + trow[i]) * a.S_pad + q0 + tcol[i], dst + OFF_DOT);
    }
    // lanes 0-31: lse[q0..+32), lanes 32-63: delta[q0..+32) -> 256 contiguous LDS bytes (one copy per wave)
    glds4((hi ? dl_row : lse_row) + q0 + l31, base + OFF_LSE + wave * 256);
  };

  f32x16_t dvacc[DO_DV ? DBLK : 1], dkacc[DO_DK ? DBLK : 1];
#pragma unroll
  for (int i = 0; i < DBLK; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if constexpr (DO_DV) dvacc[i][r] = 0.f;
      if constexpr (DO_DK) dkacc[i][r] = 0.f;
    }

  // per-lane LDS byte offsets (stage base added per tile)
  const unsigned lds0 = lds_addr_of(smem);
  unsigned off_r[KSTEPS], off_t[2];
  {
    const int r_swz = RSwz<D>::f(l31);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) off_r[ks] = l31 * (D * 2) + (((2 * ks + hi) ^ r_swz) << 4);
    const int tsw = tswz(l31);  // rows 32*i + l31: (row >> 2) & 3 does not depend on i
#pragma unroll
    for (int s = 0; s < 2; ++s) off_t[s] = l31 * 64 + (((2 * s + hi) ^ tsw) << 4);
  }
  const unsigned off_l = hi * 16 + wave * 256;
  const float sc = a.scale_log2;

#define VM_WAIT_STR2(N) "s_waitcnt vmcnt(" #N ")"
#define VM_WAIT_STR(N) VM_WAIT_STR2(N)
  // NS-1 tiles in flight; tile j is consumed after a counted wait that leaves the NS-2 newer tiles in flight
  for (int t = 0; t < NS - 1; ++t) stage(t);
  for (int j = 0; j < ntiles; ++j) {
    const int q0 = q_begin + j * 32;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr ((NS - 2) * LOADS_PER_TILE == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr ((NS - 2) * LOADS_PER_TILE == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr ((NS - 2) * LOADS_PER_TILE == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr ((NS - 2) * LOADS_PER_TILE == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr ((NS - 2) * LOADS_PER_TILE == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr ((NS - 2) * LOADS_PER_TILE == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    stage(j + NS - 1);  // refills the slot of tile j-1, which every wave finished before the barrier
    // causal: this wave's keys all above every query of the tile -> nothing to do
    if (CAUSAL && (kv0 + wave * 32 > q0 + 31)) continue;
    const unsigned sb = lds0 + (unsigned)(j % NS) * STAGE;
    unsigned ar[KSTEPS], at[2];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) ar[ks] = sb + off_r[ks];
    at[0] = sb + off_t[0];
    at[1] = sb + off_t[1];
    const unsigned al = sb + off_l;

    u32x4_t fr[NFR];
    u32x4_t lsev[4], dlv[DO_DK ? 4 : 1];
    // ---- S = Q K^T (rows = queries) ----
    static_for<KSTEPS>([&](auto I) { constexpr int ks = decltype(I)::value; lds_read128<OFF_Q>(fr[ks], ar[ks]); });
    static_for<4>([&](auto I) { constexpr int g = decltype(I)::value; lds_read128<OFF_LSE + 32 * g>(lsev[g], al); });
    if constexpr (DO_DK)
      static_for<4>([&](auto I) { constexpr int g = decltype(I)::value; lds_read128<OFF_LSE + 128 + 32 * g>(dlv[g], al); });
    LGKM_WAIT(0);
    f32x16_t sacc, pacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) sacc = mfma32v<DT>(fr[ks], kf[ks], sacc);
    if constexpr (DO_DK) {
      // ---- dP = dO V^T ----
      static_for<KSTEPS>([&](auto I) { constexpr int ks = decltype(I)::value; lds_read128<OFF_DO>(fr[ks], ar[ks]); });
      LGKM_WAIT(0);
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) pacc = mfma32v<DT>(fr[ks], vf[ks], pacc);
    }
    // ---- P and dS (query index lives in registers: q = q0 + (r&3) + 8*(r>>2) + 4*hi) ----
    float pv[16], dsv[16];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) pv[4 * g + e] = fast_exp2(fmaf(sacc[4 * g + e], sc, -__uint_as_float(lsev[g][e])));
    // masking (sequence end / diagonal) only where this wave's 32x32 block can touch it (wave-uniform branch)
    const int kvw0 = kv0 + wave * 32;
    if ((q0 + 32 > len) || (kvw0 + 32 > len) || (CAUSAL && (kvw0 + 31 > q0))) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const bool ok = (q < len) && (kvrow < len) && (!CAUSAL || kvrow <= q);
        pv[r] = ok ? pv[r] : 0.f;
      }
    }
    if constexpr (DO_DK) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e)  // softmax scale is applied once, in the epilogue
          dsv[4 * g + e] = pv[4 * g + e] * (pacc[4 * g + e] - __uint_as_float(dlv[g][e]));
    }
    if constexpr (DO_DV) {
      u32x4_t pf[2] = {pack8v<DT>(pv), pack8v<DT>(pv + 8)};
      static_for<DBLK>([&](auto I) {
        constexpr int i = decltype(I)::value;
        lds_read128<OFF_DOT + i * 2048>(fr[2 * i], at[0]);
        lds_read128<OFF_DOT + i * 2048>(fr[2 * i + 1], at[1]);
      });
      LGKM_WAIT(0);
#pragma unroll
      for (int i = 0; i < DBLK; ++i) {
        dvacc[i] = mfma32v<DT>(fr[2 * i], pf[0], dvacc[i]);
        dvacc[i] = mfma32v<DT>(fr[2 * i + 1], pf[1], dvacc[i]);
      }
    }
    if constexpr (DO_DK) {
      u32x4_t dsf[2] = {pack8v<DT>(dsv), pack8v<DT>(dsv + 8)};
      static_for<DBLK>([&](auto I) {
        constexpr int i = decltype(I)::value;
        lds_read128<OFF_QT + i * 2048>(fr[2 * i], at[0]);
        lds_read128<OFF_QT + i * 2048>(fr[2 * i + 1], at[1]);
      });
      LGKM_WAIT(0);
#pragma unroll
      for (int i = 0; i < DBLK; ++i) {
        dkacc[i] = mfma32v<DT>(fr[2 * i], dsf[0], dkacc[i]);
        dkacc[i] = mfma32v<DT>(fr[2 * i + 1], dsf[1], dkacc[i]);
      }
    }
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail re-loads must land before the LDS is released
  if (kvrow < S) {
    const bool valid = kvrow < len;
#pragma unroll
    for (int i = 0; i < DBLK; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * i + 8 * g + 4 * hi;
        if constexpr (DO_DK) {
          uint2 w = make_uint2(0, 0);
          if (valid) w = make_uint2(pack2<DT>(dkacc[i][4 * g + 0] * a.scale, dkacc[i][4 * g + 1] * a.scale), pack2<DT>(dkacc[i][4 * g + 2] * a.scale, dkacc[i][4 * g + 3] * a.scale));
          *(uint2*)(dkp + d) = w;
        }
        if constexpr (DO_DV) {
          uint2 w = make_uint2(0, 0);
          if (valid) w = make_uint2(pack2<DT>(dvacc[i][4 * g + 0], dvacc[i][4 * g + 1]), pack2<DT>(dvacc[i][4 * g + 2], dvacc[i][4 * g + 3]));
          *(uint2*)(dvp + d) = w;
        }
      }
  }
}

int g_split = 1;  // 1: MODE 1 + MODE 2 launches (default), 0: single MODE 3 launch

template <int DT, int D, bool CAUSAL>
int launch_kv(const BwdArgs& a, hipStream_t st) {
  constexpr size_t T_ = 32 * D * 2;
  constexpr size_t lds1 = 4 * (2 * T_ + 1024), lds2 = 3 * (3 * T_ + 1024), lds3 = 3 * (4 * T_ + 1024);
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)attn_bwd_kv_k<DT, D, CAUSAL, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    hipFuncSetAttribute((const void*)attn_bwd_kv_k<DT, D, CAUSAL, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    hipFuncSetAttribute((const void*)attn_bwd_kv_k<DT, D, CAUSAL, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
    attr = true;
  }
  dim3 grid(xcd_grid(a.B * a.H, (a.S + 127) / 128));
  if (g_split) {
    hipLaunchKernelGGL((attn_bwd_kv_k<DT, D, CAUSAL, 1>), grid, dim3(256), lds1, st, a);
    hipLaunchKernelGGL((attn_bwd_kv_k<DT, D, CAUSAL, 2>), grid, dim3(256), lds2, st, a);
  } else {
    hipLaunchKernelGGL((attn_bwd_kv_k<DT, D, CAUSAL, 3>), grid, dim3(256), lds3, st, a);
  }
  MH_LAUNCH_CHECK();
}

}  // namespace

int launch_attn_bwd_kv(const BwdArgs& a, int dt, int D, int causal, hipStream_t st) {
#define GO(DT_, D_, C_) return launch_kv<DT_, D_, C_>(a, st)
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

}  // namespace mhattn

extern "C" void mh_attn_bwd_split(int split) { mhattn::g_split = split; }
