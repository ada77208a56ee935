// gemm_w4: 256x256x64-tile bf16/f16 MFMA GEMM with FOUR waves per block, one 128x128 output quadrant per wave (gfx950),
// for all three operand layouts of the training step (NT forward, NN dgrad, TN wgrad; see gemm256.hip for the layouts).
//
// Why a second 256-tile kernel next to the 8-wave gemm_nt_256: the chip is power-capped (a register-only MFMA loop sustains
// ~1.7 PFLOP/s, tools/probes/mfma_probe.hip), so what a main loop reaches is set by the energy it spends per flop besides the
// MFMA itself.  With 8 waves a wave owns 128x64 outputs and a 32-deep k-step costs 12 fragment reads per 32 MFMAs; here a
// wave owns 128x128 outputs = 64 accumulator tiles = 256 fp32 registers per lane, which fits because gfx950's register file
// is 512 per lane at ONE wave per SIMD (accumulators in AccVGPRs, fragments in VGPRs): 16 fragment reads per 64 MFMAs = a
// third fewer LDS bytes per flop, one barrier pair per K-tile instead of eight, and every non-MFMA instruction of the loop
// (fragment read, M0 write, LDS-DMA copy) sits behind its own MFMA in a hand-placed stream.  Measured where it is dispatched
// (profiles/r03_gemm_w4_ab.txt): the TN weight gradients (K = 32 768 tokens: the per-tile fixed cost does not matter and the
// 8-wave kernel's transpose-read phases were its slowest form).  Since round 4 it carries EVERY product of the decoder: the store
// phase (w4_store) has one instantiation per epilogue kind - staged 16-bit store, fp32 store / accumulate, RoPE and SwiGLU on the
// fp32 accumulators, SwiGLU backward - shared with the fp8 kernel below, split-K over grid.y, K tails of K-strided operands through
// the buffer descriptor's range check, and a grouped launch for several weight gradients (profiles/r04_w4_forms_ab.txt).  The 8-wave
// kernel keeps the CLIP tower's K = 1024 products (bias / quick-GELU forms) and contractions that are not whole 128-deep K-tile pairs.
//
// LDS = two K-tile buffers of 64 KiB: [A part 32 KiB | B part 32 KiB].
//   K-contiguous part: [256 rows][128 B]; 16-byte chunk c of row r stored at c ^ ((r >> 1) & 7) (conflict-free ds_read_b128).
//   K-strided part (memory is [K][M]): two half-tiles [64 k][128 m] (256-B rows); the 32-byte column chunk is XORed with
//   f(k) = (k & 3) | ((k >> 3) & 1) << 2, fragments come from `ds_read_b64_tr_b16` transpose-reads (two per fragment).
//   Copies are LDS-DMA (`buffer_load_dwordx4 ... lds`, lane-linear LDS image => both swizzles are applied to the SOURCE
//   address); per-lane byte offsets are loop-invariant VGPRs and the K advance is the scalar offset (K-contiguous operands) or the
//   descriptor's base (K-strided operands: copy_ld): no VALU per copy.
// Per K-tile t (fragment registers double-buffered per 32-deep k-step: set 0 = k 0..31, set 1 = k 32..63):
//   phase E: 64 MFMAs on set 0 | read set 1 of tile t from buffer t&1; then lgkmcnt(0) + barrier B1 (every wave has all of
//            tile t in registers: buffer t&1 is free) and the first copies of tile t+2 into buffer t&1
//   phase O: 64 MFMAs on set 1 | vmcnt(CE) + barrier B2 (tile t+1 has landed for every wave; only the CE copies of tile t+2
//            issued in phase E may still be in flight), the remaining copies of tile t+2, read set 0 of tile t+1.
// The t loop is unrolled by the buffer parity so that every LDS address is a loop-invariant VGPR plus an immediate.
#include <type_traits>

#include "gemm_common.h"

namespace mhgemm {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#ifdef MH_W4_TIMING
// development probe (tools/probes/w4_timing.py): per block {entry, operands of tile 0 landed, main loop done, stores issued} in 100-MHz ticks + where it ran
__device__ unsigned long long* g_w4_dbg = nullptr;
__device__ __forceinline__ void w4_stamp(int k) {
  if (g_w4_dbg && threadIdx.x == 0) {
    const unsigned long long t = __builtin_amdgcn_s_memrealtime();
    g_w4_dbg[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + k] = t;
    if (k == 0) {
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      g_w4_dbg[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 4] = ((unsigned long long)xcc << 32) | hw;
    }
  }
}
#define W4_STAMP(k) w4_stamp(k)
#else
#define W4_STAMP(k)
#endif

// MFMA with the accumulator pinned to AccVGPRs ("+a"): hipcc's allocator otherwise shuttles part of the 256 accumulators
// between VGPRs and AccVGPRs inside the loop.  Volatile: the issue order below IS the schedule.
template <int DT>
__device__ __forceinline__ void mfma_acc(f32x4_t& c, const u32x4& a, const u32x4& b) {
  if constexpr (DT == MH_BF16)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

constexpr int W4_PART = 256 * 128;     // 32 KiB: 256 rows x 64 k
constexpr int W4_UNIT = 2 * W4_PART;   // A part + B part of one K-tile
constexpr int W4_CSTAGE = 128 * 272;   // per-wave C staging slice of the epilogue (128 rows x (256 + 16) B)
constexpr int W4_LDS = 4 * W4_CSTAGE;  // >= 2 * W4_UNIT: two K-tile buffers during the loop, four C slices after it
constexpr int W4_LDS_HALF = 3 * 49152; // the 128-row tile: three K-tile buffers of 48 KiB (>= the four C slices)

template <int OFF>
__device__ __forceinline__ void dsr128(u32x4& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void dsr64tr(u32x2& d, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
#define W4_FENCE() __builtin_amdgcn_sched_barrier(0)
// records of a K-contiguous operand's descriptor for tile t of nk: all of them (-1) while t < nk, none (0) past the end - in SCALAR instructions
// (written in C, `t < nk ? -1 : 0` and `(t - nk) >> 31` both come out as a VALU select whose result cannot be copied into the SGPR descriptor)
__device__ __forceinline__ int w4_nrec(int t, int nk) {
  int d;
  asm volatile("s_sub_i32 %0, %1, %2\n\ts_ashr_i32 %0, %0, 31" : "=s"(d) : "s"(t), "s"(nk) : "scc");
  return d;
}
#define W4_NREC(t, nk) w4_nrec(t, nk)
template <int N, typename F>
__device__ __forceinline__ void w4_for(F&& f) {
  if constexpr (N > 0) {
    w4_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// ---- the instruction schedule of a K-tile, as compile-time tables --------------------------------------------------------
// NR = fragment-read instructions per phase: 8 per K-contiguous operand (ds_read_b128), 16 per K-strided one (two tr reads).
template <int NR>
struct Sched {
  static constexpr int B1 = NR + 6;                                   // phase E: lgkmcnt(0) + barrier slot
  static constexpr int E0 = NR + 8;                                   // phase E: first copy slot
  static constexpr int PE = NR <= 16 ? 6 : (NR <= 24 ? 5 : 4);        // phase E: slots between copies
  static constexpr int CE = (62 - E0) / PE + 1;                       // copies issued in phase E
  static constexpr int O0 = 6, PO = 6;                                // phase O: first copy slot, spacing
  static_assert(CE >= 1 && CE <= 15 && O0 + PO * (16 - CE - 1) + 1 <= 62, "copy schedule does not fit the phases");
  static constexpr bool e_copy(int sl) { return sl >= E0 && (sl - E0) % PE < 2 && (sl - E0) / PE < CE; }
  static constexpr int e_copy_index(int sl) { return (sl - E0) / PE; }
  static constexpr bool o_copy(int sl) { return sl >= O0 && (sl - O0) % PO < 2 && (sl - O0) / PO < 16 - CE; }
  static constexpr int o_copy_index(int sl) { return CE + (sl - O0) / PO; }
  // phase O: fragment reads take the non-copy slots from 4 on, in order
  static constexpr int o_read_index(int sl, bool loads) {
    int n = 0;
    for (int s = 4; s < sl; ++s) n += (loads && o_copy(s)) ? 0 : 1;
    return n;
  }
};

// The half tile (MI = 4): 32 MFMAs per phase, 12 copies per K-tile, M0 write + copy behind the same MFMA; the fragment reads go one to a slot
// (two where there are more than 16 of them: K-strided B).
template <int NR>
struct SchedH {
  static constexpr int RPS = NR > 16 ? 2 : 1;                          // reads per slot
  static constexpr int RS = (NR + RPS - 1) / RPS;                      // slots that carry reads
  static constexpr int B1 = RS + 3;                                    // phase E: lgkmcnt(0) + barrier slot
  static constexpr int E0 = B1 + 2;                                    // phase E: first copy slot, then every second slot
  static constexpr int CE = (30 - E0) / 2 + 1 > 6 ? 6 : (30 - E0) / 2 + 1;  // copies issued in phase E
  static constexpr int O0 = 5;                                         // phase O: copy slots O0, O0 + 4, ...
  static constexpr int PO = (12 - CE) > 6 ? 3 : 4;
  static_assert(CE >= 1 && O0 + PO * (12 - CE - 1) <= 31, "copy schedule does not fit the phases");
  static constexpr bool e_copy(int sl) { return sl >= E0 && (sl - E0) % 2 == 0 && (sl - E0) / 2 < CE; }
  static constexpr int e_copy_index(int sl) { return (sl - E0) / 2; }
  static constexpr bool o_copy(int sl) { return sl >= O0 && (sl - O0) % PO == 0 && (sl - O0) / PO < 12 - CE; }
  static constexpr int o_copy_index(int sl) { return CE + (sl - O0) / PO; }
};

// Epilogue kinds (one kernel instantiation each: a run-time switch between 64-tile store blocks makes hipcc spill the accumulators
// around the merge).  The fused forms work on the fp32 ACCUMULATORS - a lane holds a rotary pair (c, c + 64) resp. a (gate, up)
// pair of one token in two accumulator tiles of its own - and round once, where the 8-wave kernel's staged forms rotate / gate the
// already rounded 16-bit tile (bit-identical to the unfused kernels; one rounding more).
enum { EK_STD = 0,         // 16-bit C through the staged store: plain / residual, optionally accumulating
       EK_F32 = 1,         // fp32 C (lm_head logits; split-K partials at C + split * c_split)
       EK_F32ACC = 2,      // fp32 C += (the fp32 residual streams: x += o Wo^T, x += act Wd^T)
       EK_ROPE = 3,        // q|k|v projection, RoPE of the q / k heads (D = 128: a wave's 128 columns are one head)
       EK_SWIGLU = 4,      // gate|up projection: a tile = 2 x (64 gate + 64 up) columns; C = gu, sw_out = act = silu(gate) * up
       EK_SWIGLU_BWD = 5 };  // dact = dY Wd -> sw_out = dgu from sw_in = gu (dact never stored)

// The store phase of both 4-wave kernels (16-bit operands: F8 = false; fp8 operands: F8 = true, where an accumulator is first multiplied by
// sc_m[row] * sc_n[column] - the per-row scales of the two quantised operands).  `acc` is the wave's 128 x 128 quadrant in accumulator registers.
template <int DT, int EK, bool F8, int MI = 8>
__device__ __forceinline__ void w4_store(const GemmArgs& g, char* smem, f32x4_t (&acc)[MI][8], int m0, int n0, int ky) {
  using std::integral_constant;
  static_assert(MI == 8 || (MI == 4 && !F8), "a wave owns MI x 8 accumulator tiles: 128 x 128 (MI = 8) or 64 x 128 (MI = 4, the 128-row block tile)");
  constexpr int RW = MI * 16;  // rows per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  float smv[F8 ? MI : 1];
  float4 snv[F8 ? 8 : 1];
  if constexpr (F8) {
#pragma unroll
    for (int i = 0; i < MI; ++i) smv[i] = g.sc_m[min(m0 + wm * RW + i * 16 + (lane & 15), g.M - 1)];
#pragma unroll
    for (int j = 0; j < 8; ++j) {  // the B row (= output column) behind accumulator tile j; SwiGLU tiles hold 64 gate + 64 up columns per wave
      const int n = EK == EK_SWIGLU ? ((j < 4 ? 0 : g.sw_ff) + n0 + wn * 64 + (j & 3) * 16 + 4 * (lane >> 4)) : (n0 + wn * 128 + j * 16 + 4 * (lane >> 4));
      snv[j] = *(const float4*)(g.sc_n + min(n, g.N - 4));
    }
  }
  auto val = [&](auto I_, auto J_, float (&v)[4]) {
    constexpr int i = decltype(I_)::value, j = decltype(J_)::value;
    v[0] = acc[i][j][0]; v[1] = acc[i][j][1]; v[2] = acc[i][j][2]; v[3] = acc[i][j][3];
    if constexpr (F8) {
      v[0] *= smv[i] * snv[j].x; v[1] *= smv[i] * snv[j].y; v[2] *= smv[i] * snv[j].z; v[3] *= smv[i] * snv[j].w;
    }
  };
  if constexpr (EK == EK_F32) {
    // fp32 C straight from the accumulators (16 bytes per lane = 64 contiguous bytes per row and instruction)
    float* cb = (float*)g.C + (int64_t)ky * g.c_split;
    w4_for<MI * 8>([&](auto T_) {
      constexpr int tt = decltype(T_)::value, i = tt / 8, j = tt % 8;
      const int m = m0 + wm * RW + i * 16 + (lane & 15);
      const int n = n0 + wn * 128 + j * 16 + 4 * (lane >> 4);
      float v[4];
      val(integral_constant<int, i>{}, integral_constant<int, j>{}, v);
      if (m < g.M && n < g.N) *(float4*)(cb + (int64_t)m * g.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
      if constexpr (tt % 8 == 7) W4_FENCE();
    });
    return;
  } else if constexpr (EK == EK_F32ACC) {
    // fp32 C += accumulators (the fp32 residual streams).  The old values do not depend on anything the block computed: they are requested
    // 16 accumulator tiles AHEAD of their use (two register sets of 16 float4 alternate), so a tile pays the memory latency once instead of
    // once per group of stores (measured per-tile fixed cost of the naive read-modify-write: 32 us against 11 us for the plain fp32 store)
    float* cb = (float*)g.C;
    float4 old[2][16];
    auto addr = [&](int tt) {  // clamped (always a valid address; the store below is guarded)
      const int i = tt / 8, j = tt % 8;
      const int m = min(m0 + wm * RW + i * 16 + (lane & 15), g.M - 1);
      const int n = min(n0 + wn * 128 + j * 16 + 4 * (lane >> 4), g.N - 4);
      return (float4*)(cb + (int64_t)m * g.ldc + n);
    };
    w4_for<16>([&](auto T_) { constexpr int t = decltype(T_)::value; old[0][t] = *addr(t); });
    W4_FENCE();
    w4_for<MI / 2>([&](auto C_) {
      constexpr int c = decltype(C_)::value;
      if constexpr (c + 1 < MI / 2) w4_for<16>([&](auto T_) { constexpr int t = decltype(T_)::value; old[(c + 1) & 1][t] = *addr((c + 1) * 16 + t); });
      W4_FENCE();
      w4_for<16>([&](auto T_) {
        constexpr int t = decltype(T_)::value, tt = c * 16 + t, i = tt / 8, j = tt % 8;
        const int m = m0 + wm * RW + i * 16 + (lane & 15);
        const int n = n0 + wn * 128 + j * 16 + 4 * (lane >> 4);
        const float4 o = old[c & 1][t];
        float v[4];
        val(integral_constant<int, i>{}, integral_constant<int, j>{}, v);
        if (m < g.M && n < g.N) *(float4*)(cb + (int64_t)m * g.ldc + n) = make_float4(v[0] + o.x, v[1] + o.y, v[2] + o.z, v[3] + o.w);
      });
      W4_FENCE();
    });
    return;
  } else {
  // Staged epilogue (16-bit C): the accumulator layout gives a lane 4 consecutive n of one row, i.e. 32-byte pieces of 16 rows
  // per store instruction.  Every wave instead packs its 128x128 quadrant into its own LDS slice ([128 rows][272 B]: 256 B of
  // data + 16 B pad, conflict-free for the 8-byte writes and the 16-byte reads) and writes it out as 16 bytes per lane = 256
  // contiguous bytes per row, 4 rows per instruction.  MH_EPI_ACCUM adds the old 16-bit values in fp32 on the way out.
  __syncthreads();  // every wave is done with the operand tiles in LDS
  W4_STAMP(5);
  char* stage = smem + wave * W4_CSTAGE;
  const unsigned st_w = lds_addr_of(stage) + (unsigned)(lane & 15) * 272 + (unsigned)(lane >> 4) * 8;
  auto stage8 = [&](const float (&v)[4], auto OFF_) {
    const uint2 pk = make_uint2(pack2<DT>(v[0], v[1]), pack2<DT>(v[2], v[3]));
    const unsigned sw_ = st_w;  // (local copy: clang rejects captured variables as asm operands in nested generic lambdas)
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(sw_), "v"(pk), "n"(decltype(OFF_)::value) : "memory");
  };
  auto fill = [&](auto EPI_) {
    constexpr int EPI = decltype(EPI_)::value;
    w4_for<MI * 8>([&](auto T_) {
      constexpr int tt = decltype(T_)::value, i = tt / 8, j = tt % 8;
      const int m = min(m0 + wm * RW + i * 16 + (lane & 15), g.M - 1);
      const int n = min(n0 + wn * 128 + j * 16 + 4 * (lane >> 4), g.N - 4);
      float v[4];
      val(integral_constant<int, i>{}, integral_constant<int, j>{}, v);
      epi_xform4<DT, EPI>(g, m, n, v);
      stage8(v, integral_constant<int, i * 16 * 272 + j * 32>{});
      if constexpr (tt % 8 == 7) W4_FENCE();  // (keeps hipcc from reading all 256 accumulators into VGPRs at once)
    });
  };
  const unsigned st_r = lds_addr_of(stage) + (unsigned)(lane >> 4) * 272 + (unsigned)(lane & 15) * 16;
  const int mrow = m0 + wm * RW + (lane >> 4);

  if constexpr (EK == EK_SWIGLU) {
    // quadrant = [64 gate | 64 up] columns n0 + 64 wn .. of ff: accumulator tiles j and j + 4 of a lane are (gate, up) of the same
    // (token, channel).  Pass 1 stages and writes gu (both halves, rounded once); pass 2 computes act = silu(gate) * up from the fp32
    // accumulators, stages its 64 columns in the same slice and writes them.
    fill(integral_constant<int, 0>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int cq = lane & 15, gcol = n0 + wn * 64 + (cq & 7) * 8;  // channel (< ff) of this lane's 16-byte chunk
    uint16_t* gp = (uint16_t*)g.C + (int64_t)mrow * g.ldc + (cq < 8 ? 0 : g.sw_ff) + gcol;
    const bool c_ok = gcol < g.sw_ff;
#pragma unroll
    for (int part = 0; part < MI / 2; ++part) {
      u32x4 rv[8];
      w4_for<8>([&](auto R_) { constexpr int r = decltype(R_)::value; dsr128<r * 4 * 272>(rv[r], st_r + (unsigned)part * 32 * 272); });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      W4_FENCE();
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int row = part * 32 + r * 4;
        if (c_ok && mrow + row < g.M) *(u32x4*)(gp + (int64_t)row * g.ldc) = rv[r];
      }
    }
    w4_for<MI * 4>([&](auto T_) {
      constexpr int tt = decltype(T_)::value, i = tt / 4, j = tt % 4;
      float v[4], gt[4], up[4];
      val(integral_constant<int, i>{}, integral_constant<int, j>{}, gt);
      val(integral_constant<int, i>{}, integral_constant<int, j + 4>{}, up);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = swiglu_fwd1(gt[e], up[e]);
      stage8(v, integral_constant<int, i * 16 * 272 + j * 32>{});
      if constexpr (tt % 4 == 3) W4_FENCE();
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned st_r2 = lds_addr_of(stage) + (unsigned)(lane >> 3) * 272 + (unsigned)(lane & 7) * 16;
    const int arow = m0 + wm * RW + (lane >> 3), acol = n0 + wn * 64 + (lane & 7) * 8;
    uint16_t* ap = (uint16_t*)g.sw_out + (int64_t)arow * g.sw_ldo + acol;
#pragma unroll
    for (int part = 0; part < MI / 4; ++part) {
      u32x4 rv[8];
      w4_for<8>([&](auto R_) { constexpr int r = decltype(R_)::value; dsr128<r * 8 * 272>(rv[r], st_r2 + (unsigned)part * 64 * 272); });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      W4_FENCE();
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int row = part * 64 + r * 8;
        if (acol < g.sw_ff && arow + row < g.M) *(u32x4*)(ap + (int64_t)row * g.sw_ldo) = rv[r];
      }
    }
    return;
  } else if constexpr (EK == EK_ROPE) {
    // the wave's 128 columns are ONE head (n0 % 256 == 0, D = 128): channel c < 64 sits in accumulator tile j = c / 16, its rotary
    // partner c + 64 in tile j + 4 of the same lane and row -> rotate-half RoPE on the fp32 accumulators, one rounding
    // (the v heads behind rope_cols take the same block with (cos, sin) = (1, 0): x * 1 - y * 0 is exact, and ONE store block avoids the
    // accumulator spills hipcc produces around a merge of two)
    const bool rot = n0 + wn * 128 < g.rope_cols;
    // (cos, sin) of row group i + 1 are requested while group i is rotated and staged (two register sets of 8 float4): the table reads
    // (L2-resident, but ~1 us away) are paid once per tile instead of once per row group
    float4 tb[2][8];
    auto fetch = [&](auto I_) {
      constexpr int i = decltype(I_)::value;
      const int m = min(m0 + wm * RW + i * 16 + (lane & 15), g.M - 1);
      const float4* t4 = (const float4*)(g.rope_tab + ((int64_t)(m % g.rope_S) * 64 + 4 * (lane >> 4)) * 2);
      w4_for<4>([&](auto J_) {
        constexpr int j = decltype(J_)::value;
        tb[i & 1][2 * j] = t4[8 * j];          // channels j*16 + 4*(lane>>4) + {0, 1}: (c0, s0, c1, s1)
        tb[i & 1][2 * j + 1] = t4[8 * j + 1];  // + {2, 3}
      });
    };
    fetch(integral_constant<int, 0>{});
    W4_FENCE();
    w4_for<MI>([&](auto I_) {
      constexpr int i = decltype(I_)::value;
      if constexpr (i + 1 < MI) fetch(integral_constant<int, i + 1>{});
      w4_for<4>([&](auto J_) {
        constexpr int j = decltype(J_)::value;
        const float4 t01 = tb[i & 1][2 * j], t23 = tb[i & 1][2 * j + 1];
        const float cs[4] = {rot ? t01.x : 1.f, rot ? t01.z : 1.f, rot ? t23.x : 1.f, rot ? t23.z : 1.f};
        const float sn[4] = {rot ? t01.y : 0.f, rot ? t01.w : 0.f, rot ? t23.y : 0.f, rot ? t23.w : 0.f};
        float lo[4], hi[4], xa[4], xb[4];
        val(integral_constant<int, i>{}, integral_constant<int, j>{}, xa);
        val(integral_constant<int, i>{}, integral_constant<int, j + 4>{}, xb);
#pragma unroll
        for (int e = 0; e < 4; ++e) rope_rot(xa[e], xb[e], cs[e], sn[e], lo[e], hi[e]);
        stage8(lo, integral_constant<int, i * 16 * 272 + j * 32>{});
        stage8(hi, integral_constant<int, i * 16 * 272 + (j + 4) * 32>{});
      });
      W4_FENCE();
    });
  } else if constexpr (EK == EK_SWIGLU_BWD) {
    fill(integral_constant<int, 0>{});  // dact, rounded to 16 bits as the unfused path stores it
  } else {
    // (two variants only: every further instantiation of this 64-tile block behind a switch makes hipcc spill more of the accumulators
    // around the merge - the bias / quick-GELU epilogues belong to the CLIP tower's K = 1024 GEMMs, which stay on the 8-wave kernel anyway)
    if ((g.epi & ~MH_EPI_ACCUM) == MH_EPI_RESIDUAL) fill(integral_constant<int, MH_EPI_RESIDUAL>{});
    else fill(integral_constant<int, 0>{});
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int ncol = n0 + wn * 128 + (lane & 15) * 8;
  const bool n_ok = ncol < g.N;
  if constexpr (EK == EK_SWIGLU_BWD) {
    // dgu[m, n] , dgu[m, ff + n] from gu[m, n], gu[m, ff + n] and the staged dact chunk: 16-byte row pieces in and out.  The gu pieces of the
    // NEXT 32-row part are requested while the current part is computed and stored (two register sets): their latency is paid once per
    // tile, behind the LDS round trip of the first part, instead of once per part.
    const int ncl = min(ncol, g.N - 8);  // (clamped: always a valid address; stores are guarded)
    const uint16_t* gup = (const uint16_t*)g.sw_in + ncl;
    uint16_t* dgp = (uint16_t*)g.sw_out + (int64_t)mrow * g.sw_ldo + ncol;
    uint4 gq[2][8], uq[2][8];
    // fp8 step: maxima of the stored |dgu| for the quantisers behind this GEMM (GemmArgs::amax_r / amax_c): per lane 8 gate and 8 up columns
    float cmg[F8 ? 8 : 1], cmu[F8 ? 8 : 1];
    const bool want_amax = F8 && g.amax_r != nullptr;
    if constexpr (F8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) cmg[k] = cmu[k] = 0.f;
    }
    auto fetch = [&](auto P_) {
      constexpr int part = decltype(P_)::value;
      w4_for<8>([&](auto R_) {
        constexpr int r = decltype(R_)::value;
        const uint16_t* src = gup + (int64_t)min(mrow + part * 32 + r * 4, g.M - 1) * g.sw_ldi;
        gq[part & 1][r] = *(const uint4*)src;
        uq[part & 1][r] = *(const uint4*)(src + g.sw_ff);
      });
    };
    fetch(integral_constant<int, 0>{});
    W4_FENCE();
    w4_for<MI / 2>([&](auto P_) {
      constexpr int part = decltype(P_)::value;
      u32x4 rv[8];
      w4_for<8>([&](auto R_) { constexpr int r = decltype(R_)::value; dsr128<r * 4 * 272>(rv[r], st_r + (unsigned)part * 32 * 272); });
      if constexpr (part + 1 < MI / 2) fetch(integral_constant<int, part + 1>{});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      W4_FENCE();
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int row = part * 32 + r * 4;
        float d_[8], ga[8], ub[8], dg[8], du[8];
        unpack8<DT>(uint4{rv[r][0], rv[r][1], rv[r][2], rv[r][3]}, d_);
        unpack8<DT>(gq[part & 1][r], ga);
        unpack8<DT>(uq[part & 1][r], ub);
#pragma unroll
        for (int k = 0; k < 8; ++k) swiglu_bwd1(ga[k], ub[k], d_[k], dg[k], du[k]);
        const uint4 pg = pack8<DT>(dg), pu = pack8<DT>(du);
        const bool ok = n_ok && mrow + row < g.M;
        if (ok) {
          *(uint4*)(dgp + (int64_t)row * g.sw_ldo) = pg;
          *(uint4*)(dgp + (int64_t)row * g.sw_ldo + g.sw_ff) = pu;
        }
        if constexpr (F8) {
          if (want_amax) {  // (wave-uniform)
            float rg[8], ru[8], rm = 0.f;
            unpack8<DT>(pg, rg);
            unpack8<DT>(pu, ru);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const float a = ok ? fabsf(rg[k]) : 0.f, b = ok ? fabsf(ru[k]) : 0.f;
              rm = fmaxf(rm, fmaxf(a, b));
              cmg[k] = fmaxf(cmg[k], a);
              cmu[k] = fmaxf(cmu[k], b);
            }
            // the 16 lanes (lane & 15) of one lane >> 4 hold the 128 gate + 128 up columns of this row
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) rm = fmaxf(rm, __shfl_xor(rm, o, 64));
            if ((lane & 15) == 0 && mrow + row < g.M && rm > 0.f) atomicMax(g.amax_r + mrow + row, __float_as_uint(rm));
          }
        }
      }
      W4_FENCE();
    });
    if constexpr (F8) {
      if (want_amax) {  // column maxima over the wave's 128 rows: the four lane >> 4 groups hold different rows of the same columns
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          cmg[k] = fmaxf(cmg[k], __shfl_xor(cmg[k], 16, 64)); cmg[k] = fmaxf(cmg[k], __shfl_xor(cmg[k], 32, 64));
          cmu[k] = fmaxf(cmu[k], __shfl_xor(cmu[k], 16, 64)); cmu[k] = fmaxf(cmu[k], __shfl_xor(cmu[k], 32, 64));
        }
        if (lane < 16 && n_ok) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            if (cmg[k] > 0.f) atomicMax(g.amax_c + ncol + k, __float_as_uint(cmg[k]));
            if (cmu[k] > 0.f) atomicMax(g.amax_c + g.sw_ff + ncol + k, __float_as_uint(cmu[k]));
          }
        }
      }
    }
    return;
  }
  uint16_t* cp = (uint16_t*)g.C + (int64_t)mrow * g.ldc + ncol;
  W4_STAMP(6);
  const bool accum = EK == EK_STD && (g.epi & MH_EPI_ACCUM) != 0;
#pragma unroll
  for (int part = 0; part < MI / 2; ++part) {
    u32x4 rv[8];
    w4_for<8>([&](auto R_) { constexpr int r = decltype(R_)::value; dsr128<r * 4 * 272>(rv[r], st_r + (unsigned)part * 32 * 272); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W4_FENCE();
    if (accum) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int row = part * 32 + r * 4;
        if (n_ok && mrow + row < g.M) {
          const uint4 old = *(const uint4*)(cp + (int64_t)row * g.ldc);
          float a[8], o[8];
          unpack8<DT>(uint4{rv[r][0], rv[r][1], rv[r][2], rv[r][3]}, a);
          unpack8<DT>(old, o);
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] += o[e];
          *(uint4*)(cp + (int64_t)row * g.ldc) = pack8<DT>(a);
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int row = part * 32 + r * 4;
        if (n_ok && mrow + row < g.M) *(u32x4*)(cp + (int64_t)row * g.ldc) = rv[r];
      }
    }
  }
  }
}

// max(0, span - off) in scalar arithmetic.  (Written as `off < span ? span - off : 0` or `span - min(off, span)` hipcc recognises an
// unsigned saturating subtract, which exists only as a VALU instruction (v_sub_u32 clamp): the descriptor word would have to come back
// through a VGPR -> SGPR copy the backend refuses.)
__device__ __forceinline__ unsigned left_after(unsigned span, unsigned off) {
  const long long d = (long long)span - (long long)off;
  return d > 0 ? (unsigned)d : 0u;
}

// MI = accumulator tiles of a wave along M: 8 = the 256 x 256 block tile described above; 4 = a 128 x 256 block tile (a wave owns 64 x 128; the A part
// of a K-tile is 128 rows = 4 copies per wave, 12 copies and 32 MFMAs per phase) for products with few rows: a 613-token prefill is 2.4 tiles of
// 256 rows - its gate|up product is 3 x 86 = 258 tiles = two rounds of the 256 CUs for 1.008 rounds of work and q|k|v fills 144 of them; in half
// tiles they are 430 = 1.7 and 240 = 0.94 half-rounds (profiles/r05_cfg2_fwd_kernel_stats.txt; the host chooses, gemm.hip).
template <int DT, bool AKS, bool BKS, int EK, int MI = 8>
__device__ __forceinline__ void w4_tile(const GemmArgs& g, char* smem, int tm, int tn, int ky) {
  static_assert(MI == 8 || MI == 4, "MI");
  constexpr int RW = MI * 16;      // rows of A per wave
  constexpr int NCOPY = MI + 8;    // LDS-DMA copies per wave and K-tile: MI of the A part, 8 of the B part
  constexpr int NS = MI * 8;       // MFMAs per phase
  // K-tile buffers in LDS.  The full tile has two of 64 KiB.  A half tile's K-tile is 64 MFMAs per wave = ~0.5 us, so copies requested one
  // tile-and-a-half ahead would have to land inside ~0.7 us - less than an HBM round trip under load (measured: 1.0 us per K-tile with two
  // buffers, twice its MFMA time): its buffers are packed to 48 KiB (A 16 | B 32) and THREE of them fit, copies run two-and-a-half tiles ahead.
  constexpr int NBUF = MI == 8 ? 2 : 3;
  constexpr int BOFF = MI == 8 ? W4_PART : 16384;       // B part of a buffer
  constexpr int UNIT = MI == 8 ? W4_UNIT : 49152;       // bytes per buffer
  static_assert(MI == 8 || (!AKS && !BKS), "the half tile is instantiated for NT products (tiles past the end of a K-strided operand's SPLIT are not zeros)");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = tm * (2 * RW), n0 = tn * (EK == EK_SWIGLU ? 128 : 256);
  // K range of this block: all of K, or split `ky` of g.splits (an even number of K-tiles each; K-strided operands may end inside
  // a tile - rows k >= K read as zeros through the buffer descriptor's range check, see copy_ld)
  const int nkt = (g.K + 2 * BK - 1) / (2 * BK) * 2;
  const int per = g.splits > 1 ? ((nkt / 2 + g.splits - 1) / g.splits) * 2 : nkt;
  const int kt0 = ky * per;
  const int nk = min(per, nkt - kt0);
  constexpr int NRA = AKS ? 2 * MI : MI, NRB = BKS ? 16 : 8, NR = NRA + NRB;
  using SC = std::conditional_t<MI == 8, Sched<NR>, SchedH<NR>>;

  // ---- copies: per-lane source byte offsets (loop-invariant) and the wave's LDS destinations --------------------------------
  // K-contiguous part: wave-load j (0..7) covers part rows 64*wave + 8j .. +7; lane i -> row + (i>>3), physical chunk i&7.
  // K-strided part:    wave-load j covers half j>>2, 16-byte chunks qd = (j&3)*256 + 64*wave + lane of its [64 k][128 m] image.
  int voffA[8], voffB[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if constexpr (!AKS) {
      const int row = wave * (MI * 8) + (j & (MI - 1)) * 8 + (lane >> 3);  // (MI = 4: entries 4..7 repeat 0..3 and are never copied)
      voffA[j] = (int)((int64_t)min(m0 + row, g.M - 1) * g.lda * 2 + ((lane & 7) ^ ((row >> 1) & 7)) * 16);
    } else {
      const int qd = (j & 3) * 256 + wave * 64 + lane, k = qd >> 4, cc = qd & 15;
      const int col = ((((cc >> 1) ^ ((k & 3) | (((k >> 3) & 1) << 2))) << 1) | (cc & 1)) * 8;
      voffA[j] = (int)(((int64_t)k * g.lda + min(m0 + (MI == 8 ? (j >> 2) * 128 : 0) + col, g.M - 8)) * 2);
    }
    if constexpr (!BKS) {
      const int row = wave * 64 + j * 8 + (lane >> 3);
      int src = n0 + row;
      if constexpr (EK == EK_SWIGLU) {  // part rows [64 q, 64 q + 64): q = 2 * (column half) + (0 gate | 1 up)
        const int q = row >> 6;
        src = (q & 1) * g.sw_ff + n0 + (q >> 1) * 64 + (row & 63);
      }
      voffB[j] = (int)((int64_t)min(src, g.N - 1) * g.ldb * 2 + ((lane & 7) ^ ((row >> 1) & 7)) * 16);
    } else {
      const int qd = (j & 3) * 256 + wave * 64 + lane, k = qd >> 4, cc = qd & 15;
      const int col = ((((cc >> 1) ^ ((k & 3) | (((k >> 3) & 1) << 2))) << 1) | (cc & 1)) * 8;
      voffB[j] = (int)(((int64_t)k * g.ldb + min(n0 + (j >> 2) * 128 + col, g.N - 8)) * 2);
    }
  }
#if defined(W4H_EXPERIMENT) && W4H_EXPERIMENT == 5  // timing probe: K-contiguous operands fetched as if stored TILE-MAJOR (1 KiB contiguous per copy instruction, 32 KiB per tile and K-tile); results are garbage
  if constexpr (MI == 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (!AKS) voffA[j] = (m0 / 256) * (g.K / 64) * 32768 + wave * 8192 + j * 1024 + lane * 16;
      if constexpr (!BKS) voffB[j] = (n0 / 256) * (g.K / 64) * 32768 + wave * 8192 + j * 1024 + lane * 16;
    }
  }
#define W4_KC_STEP 32768u
#else
#define W4_KC_STEP (unsigned)(BK * 2)
#endif
  const unsigned kstepA = AKS ? (unsigned)((int64_t)BK * g.lda * 2) : BK * 2;  // source bytes per K-tile (host guarantees K * ld * 2 < 2^32)
  const unsigned kstepB = BKS ? (unsigned)((int64_t)BK * g.ldb * 2) : BK * 2;
  // operand bases at this block's first K-tile (wave-uniform) and, for K-strided operands, the bytes from there to the end of row K - 1
  const uint64_t baseA = (uint64_t)(uintptr_t)g.A + (uint64_t)kt0 * kstepA, baseB = (uint64_t)(uintptr_t)g.B + (uint64_t)kt0 * kstepB;
  const unsigned spanA = AKS ? (unsigned)((int64_t)(g.K - kt0 * BK) * g.lda * 2) : 0u;
  const unsigned spanB = BKS ? (unsigned)((int64_t)(g.K - kt0 * BK) * g.ldb * 2) : 0u;
  auto make_rs = [](uint64_t a_, unsigned nrec) {
    return i32x4{__builtin_amdgcn_readfirstlane((int)(uint32_t)a_),
                 __builtin_amdgcn_readfirstlane((int)(uint32_t)((a_ >> 32) & 0xffffu)), __builtin_amdgcn_readfirstlane((int)nrec), 0x00020000};
  };
  const i32x4 rsA = make_rs(baseA, 0xffffffffu), rsB = make_rs(baseB, 0xffffffffu);
  const unsigned lds0 = lds_addr_of(smem);
  // the wave's share of a part starts at wave * 8 KiB (K-contiguous: 64 rows) or wave * 1 KiB inside each 4-KiB group (K-strided)
  const unsigned w_kc = lds0 + (unsigned)wave * 8192u, w_ks = lds0 + (unsigned)wave * 1024u;  // (wave-uniform: SGPRs)
  const unsigned w_kca = lds0 + (unsigned)wave * (unsigned)(MI * 1024);  // A part, K-contiguous: MI * 8 rows per wave
  // the M0 write and the copy are separate single instructions, each placed behind its own MFMA
  auto copy_m0 = [&](auto C_, auto BUF_) {
    constexpr int c = decltype(C_)::value, bu = decltype(BUF_)::value, j = c & 7;
    constexpr bool ks = c < 8 ? AKS : BKS;
    constexpr int imm = bu * UNIT + (c >> 3) * BOFF + (ks ? (j >> 2) * 16384 + (j & 3) * 4096 : j * 1024);
    const unsigned base_ = ks ? w_ks : (c < 8 ? w_kca : w_kc);  // (local copy: clang rejects captured variables as asm operands in nested generic lambdas)
    asm volatile("s_add_u32 m0, %0, %1" ::"s"(base_), "n"(imm) : "scc");
  };
  // K-contiguous operand: fixed descriptor, the K advance is the scalar offset.  K-strided operand: the K advance goes into the
  // descriptor's BASE and its range shrinks to the bytes left up to row K - 1 (a handful of scalar instructions per K-tile), so that
  // rows k >= K of the last tile(s) fail the hardware range check and arrive in LDS as zeros: K need not be a multiple of the K-tile
  // (the CLIP tower's 27 696 tokens) - the scalar offset of a buffer instruction takes no part in that check, the base does.
  // The descriptors of tile t are built from scalar values only (kernel arguments, block ids, the loop counter) and are pinned in SGPRs
  // by tile_rs() at the START of a phase, many instructions ahead of the copies that read them: the copies are inline asm, so the
  // compiler's hazard recogniser cannot see that a VMEM instruction reads those SGPRs, and a descriptor word written by a VALU
  // instruction (v_readfirstlane) just before it would be read stale.
  struct TileRs { i32x4 a, b; unsigned sa, sb; };
  auto tile_rs = [&](int t) {
    TileRs r;
    if constexpr (AKS) {
      const unsigned off = (unsigned)t * kstepA;
      const uint64_t ba = baseA + off;
      r.a = i32x4{(int)(uint32_t)ba, (int)(uint32_t)((ba >> 32) & 0xffffu), (int)left_after(spanA, off), 0x00020000};
      r.sa = 0;
    } else {
      if constexpr (NBUF == 3) r.a = i32x4{rsA[0], rsA[1], W4_NREC(t, nk), 0x00020000};  // (three buffers: a tile past the end is copied as ZEROS - no records - and multiplied like any other)
      else r.a = rsA;
      r.sa = (unsigned)t * W4_KC_STEP;
    }
    if constexpr (BKS) {
      const unsigned off = (unsigned)t * kstepB;
      const uint64_t bb = baseB + off;
      r.b = i32x4{(int)(uint32_t)bb, (int)(uint32_t)((bb >> 32) & 0xffffu), (int)left_after(spanB, off), 0x00020000};
      r.sb = 0;
    } else {
      if constexpr (NBUF == 3) r.b = i32x4{rsB[0], rsB[1], W4_NREC(t, nk), 0x00020000};  // (-1 = every record while t < nk, else 0; scalar arithmetic - a select becomes a VALU instruction here)
      else r.b = rsB;
      r.sb = (unsigned)t * W4_KC_STEP;
    }
    asm volatile("" : "+s"(r.a), "+s"(r.b), "+s"(r.sa), "+s"(r.sb));
    return r;
  };
  bool in_loop = false;  // (development probes only)
  auto copy_ld = [&](auto C_, const TileRs& r) {
    constexpr int c = decltype(C_)::value;
    constexpr bool isA = c < 8;
    const int vo = isA ? voffA[c & 7] : voffB[c & 7];
    const i32x4 rs = isA ? r.a : r.b;
    const unsigned soff = isA ? r.sa : r.sb;
#if defined(W4H_EXPERIMENT)  // development probes of the main loop (profiles/r05_w4_fetch_probe.txt): 1 / 3 = no copies inside the loop (half tile / every tile), 2 / 4 = the same requests into registers
    if (in_loop && ((MI == 4 && W4H_EXPERIMENT <= 2) || W4H_EXPERIMENT >= 3)) {
      if (W4H_EXPERIMENT == 1 || W4H_EXPERIMENT == 3) return;
      u32x4 sink;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(sink) : "v"(vo), "s"(rs), "s"(soff) : "memory");
      return;
    }
#endif
#ifndef W4_COPY_MOD
#define W4_COPY_MOD ""   // (cache-policy bits of the tile copies, a development switch: " nt", " sc1", " sc0 sc1"; measured: profiles/r05_w4_fetch_probe.txt)
#endif
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen" W4_COPY_MOD " lds" ::"v"(vo), "s"(rs), "s"(soff) : "memory");
  };
  // copy k of a K-tile (0 .. NCOPY-1) -> copy slot c: 0 .. MI-1 = the A part, 8 .. 15 = the B part
  auto copy_m0k = [&](auto K_, auto BUF_) { constexpr int k = decltype(K_)::value; copy_m0(std::integral_constant<int, (k < MI ? k : 8 + k - MI)>{}, BUF_); };
  auto copy_ldk = [&](auto K_, const TileRs& r) { constexpr int k = decltype(K_)::value; copy_ld(std::integral_constant<int, (k < MI ? k : 8 + k - MI)>{}, r); };
  auto issue_tile = [&](auto BUF_, int t) {  // prologue form (all copies back to back)
    const TileRs r = tile_rs(t);
    w4_for<NCOPY>([&](auto K_) {
      copy_m0k(K_, BUF_);
      copy_ldk(K_, r);
    });
  };

  using std::integral_constant;
  f32x4_t acc[MI][8];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- fragment addresses (buffer 0 and buffer 1: loop-invariant VGPRs, everything else is an immediate) --------------------
  const int fr = lane & 15, kq = lane >> 4;
  const unsigned swz = (unsigned)((fr >> 1) & 7);
  // K-contiguous: lane row inside the wave's 128 rows, k-step s chunk (4s + kq) ^ swz; fragment q = rows 16q.. = +2048 q
  unsigned a_kc[NBUF][2], b_kc[NBUF][2];   // [buffer][k-step]
  // K-strided: lane points at row k = 8*kq + (fr>>2), columns 4*(fr&3)..+3 of fragment q's 32-byte chunk (q ^ fx)
  unsigned a_ks[NBUF][MI], b_ks[NBUF][8];  // [buffer][fragment]
  {
    const unsigned t_rel = (unsigned)(kq * 8 + (fr >> 2)) * 256 + (unsigned)(fr & 3) * 8;
    const unsigned fx = (unsigned)(fr >> 2) | ((unsigned)(kq & 1) << 2);
#pragma unroll
    for (int b = 0; b < NBUF; ++b) {
      const unsigned ub = lds0 + (unsigned)b * UNIT;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        a_kc[b][s] = ub + (unsigned)(wm * RW + fr) * 128 + (((4 * s + kq) ^ swz) << 4);
        b_kc[b][s] = ub + BOFF + (unsigned)(wn * 128 + fr) * 128 + (((4 * s + kq) ^ swz) << 4);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        // MI = 8: a wave's 128 m are one [64 k][128 m] half-tile; MI = 4: its 64 m are the 32-byte chunks 4 wm .. 4 wm + 3 of the only one
        if (q < MI) a_ks[b][q] = MI == 8 ? ub + (unsigned)wm * 16384 + t_rel + (((unsigned)q ^ fx) << 5) : ub + t_rel + (((unsigned)(wm * 4 + q) ^ fx) << 5);
        b_ks[b][q] = ub + BOFF + (unsigned)wn * 16384 + t_rel + (((unsigned)q ^ fx) << 5);
      }
    }
  }

  u32x4 afc[2][MI], bfc[2][8];                           // K-contiguous fragments [set][fragment]
  u32x2 afl[2][MI], afh[2][MI], bfl[2][8], bfh[2][8];    // K-strided fragments: k 0..3 | 4..7 of the lane's 8 (two tr reads)

    // fragment-read instruction r (0..NR-1) of k-step SET from buffer BUF: A's reads first, then B's
  auto read1 = [&](auto BUF_, auto SET_, auto R_) {
    constexpr int bu = decltype(BUF_)::value, s = decltype(SET_)::value, r = decltype(R_)::value;
    if constexpr (r < NRA) {
      if constexpr (!AKS) {
        dsr128<r * 2048>(afc[s][r], a_kc[bu][s]);
      } else {
        constexpr int q = r >> 1;
        if constexpr ((r & 1) == 0) dsr64tr<s * 8192>(afl[s][q], a_ks[bu][q]);
        else dsr64tr<s * 8192 + 1024>(afh[s][q], a_ks[bu][q]);
      }
    } else {
      constexpr int rb = r - NRA;
      if constexpr (!BKS) {
        dsr128<rb * 2048>(bfc[s][rb], b_kc[bu][s]);
      } else {
        constexpr int q = rb >> 1;
        if constexpr ((rb & 1) == 0) dsr64tr<s * 8192>(bfl[s][q], b_ks[bu][q]);
        else dsr64tr<s * 8192 + 1024>(bfh[s][q], b_ks[bu][q]);
      }
    }
  };
  // MFMA slot sl (0..NS-1): accumulator (sl % MI, sl / MI).  The matrix core takes an independent 16x16x32 MFMA every 16 cycles and a
  // wave issues in order, so with ONE wave per SIMD at most one other instruction may sit between two MFMAs.
  auto mfma_slot = [&](auto SET_, auto SL_) {
    constexpr int s = decltype(SET_)::value, sl = decltype(SL_)::value, i = sl % MI, j = sl / MI;
    u32x4 a, b;
    if constexpr (AKS) a = u32x4{afl[s][i][0], afl[s][i][1], afh[s][i][0], afh[s][i][1]};
    else a = afc[s][i];
    if constexpr (BKS) b = u32x4{bfl[s][j][0], bfl[s][j][1], bfh[s][j][0], bfh[s][j][1]};
    else b = bfc[s][j];
    mfma_acc<DT>(acc[i][j], b, a);  // operands swapped: the accumulator holds 4 consecutive n of one row m (vector epilogue)
  };
  using S0 = integral_constant<int, 0>;
  using S1 = integral_constant<int, 1>;

  // (tiles past the end: the copies re-fetch the LAST tile into a buffer nobody reads again - one uniform loop body, no tail variants:
  // hipcc spills hundreds of registers around control flow that merges paths through these hand-placed asm streams)
  auto phase_e = [&](auto BUF_, int t) {  // MFMAs on set 0 (tile t, k 0..31); fetch set 1 of tile t
    constexpr bool loads = true;
    const TileRs tr = tile_rs(NBUF == 3 ? t + NBUF : min(t + NBUF, nk - 1));
    w4_for<NS>([&](auto SL_) {
      constexpr int sl = decltype(SL_)::value;
      mfma_slot(S0{}, SL_);
      if constexpr (MI == 8) {
        if constexpr (sl < NR) read1(BUF_, S1{}, integral_constant<int, sl>{});
      } else {  // half tile: the reads go two to a slot where there are more than 16 of them
        constexpr int per = SC::RPS;
        if constexpr (sl * per < NR) read1(BUF_, S1{}, integral_constant<int, sl * per>{});
        if constexpr (per == 2 && sl * per + 1 < NR) read1(BUF_, S1{}, integral_constant<int, sl * per + 1>{});
      }
      if constexpr (loads && sl == SC::B1) {  // every wave has all of tile t in registers -> buffer bu is free
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      if constexpr (MI == 8) {
        if constexpr (loads && SC::e_copy(sl)) {
          if constexpr ((sl - SC::E0) % SC::PE == 0) copy_m0(integral_constant<int, SC::e_copy_index(sl)>{}, BUF_);
          else copy_ld(integral_constant<int, SC::e_copy_index(sl)>{}, tr);
        }
      } else if constexpr (loads && SC::e_copy(sl)) {  // (M0 write and copy behind the same MFMA)
        copy_m0k(integral_constant<int, SC::e_copy_index(sl)>{}, BUF_);
        copy_ldk(integral_constant<int, SC::e_copy_index(sl)>{}, tr);
      }
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W4_FENCE();
  };
  auto phase_o = [&](auto BUF_, int t) {  // MFMAs on set 1 (tile t, k 32..63); fetch set 0 of tile t+1
    constexpr int bu = decltype(BUF_)::value;
    constexpr bool loads = true;
    const TileRs tr = tile_rs(NBUF == 3 ? t + NBUF : min(t + NBUF, nk - 1));
    using NB = integral_constant<int, (bu + 1) % NBUF>;
    w4_for<NS>([&](auto SL_) {
      constexpr int sl = decltype(SL_)::value;
      mfma_slot(S1{}, SL_);
      if constexpr (sl == 3) {  // tile t+1 has landed for every wave (only phase E's copies of tile t+NBUF - and all of the tiles between - may still be in flight)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * NCOPY + SC::CE) : "memory");
        __builtin_amdgcn_s_barrier();
      }
      constexpr bool cp = loads && SC::o_copy(sl);
      if constexpr (MI == 8) {
        if constexpr (cp) {
          if constexpr ((sl - SC::O0) % SC::PO == 0) copy_m0(integral_constant<int, SC::o_copy_index(sl)>{}, BUF_);
          else copy_ld(integral_constant<int, SC::o_copy_index(sl)>{}, tr);
        }
        constexpr int nread = SC::o_read_index(sl, loads);
        if constexpr (!cp && sl >= 4 && nread < NR) read1(NB{}, S0{}, integral_constant<int, nread>{});  // (past the last tile: dead data, unused)
      } else {
        if constexpr (cp) {
          copy_m0k(integral_constant<int, SC::o_copy_index(sl)>{}, BUF_);
          copy_ldk(integral_constant<int, SC::o_copy_index(sl)>{}, tr);
        }
        constexpr int per = SC::RPS, r0 = (sl - 4) * per;
        if constexpr (sl >= 4 && r0 < NR) read1(NB{}, S0{}, integral_constant<int, (r0 < NR ? r0 : 0)>{});
        if constexpr (sl >= 4 && per == 2 && r0 + 1 < NR) read1(NB{}, S0{}, integral_constant<int, (r0 + 1 < NR ? r0 + 1 : 0)>{});
      }
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W4_FENCE();
  };

  // prologue: tiles 0 .. NBUF-1 in flight, tile 0 landed, its k-step 0 fragments in registers
  issue_tile(integral_constant<int, 0>{}, 0);
  issue_tile(integral_constant<int, 1>{}, 1);
  if constexpr (NBUF == 3) issue_tile(integral_constant<int, 2>{}, 2);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 1) * NCOPY) : "memory");
  __builtin_amdgcn_s_barrier();
  W4_FENCE();
  w4_for<NR>([&](auto R_) { read1(S0{}, S0{}, R_); });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  W4_FENCE();
  W4_STAMP(1);

  in_loop = true;
  if constexpr (NBUF == 2) {
    for (int t = 0; t < nk; t += 2) {  // nk is even: tile t from buffer 0, tile t+1 from buffer 1
      phase_e(S0{}, t);
      phase_o(S0{}, t);
      phase_e(S1{}, t + 1);
      phase_o(S1{}, t + 1);
    }
  } else {
    using S2 = integral_constant<int, 2>;
    // tile t from buffer t % 3, whole groups of three: the one or two tiles past the end of K are zeros (tile_rs) - exits from the middle of
    // the group made hipcc merge three register assignments of the accumulators (256 AGPRs, 20 spills; the straight loop: 128, none)
    for (int t = 0; t < nk; t += 3) {
      phase_e(S0{}, t);
      phase_o(S0{}, t);
      phase_e(S1{}, t + 1);
      phase_o(S1{}, t + 1);
      phase_e(S2{}, t + 2);
      phase_o(S2{}, t + 2);
    }
  }
  // the s_nops cover the MFMA -> accumulator-read hazard that the compiler cannot see through the inline-asm MFMAs
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  W4_STAMP(2);

  w4_store<DT, EK, false, MI>(g, smem, acc, m0, n0, ky);
  W4_STAMP(3);
}

template <int DT, bool AKS, bool BKS, int EK, int MI = 8>
__global__ __launch_bounds__(256, 1) void gemm_w4(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  W4_STAMP(0);
  int tm, tn;
  tile_of_block(g, tm, tn);
  w4_tile<DT, AKS, BKS, EK, MI>(g, smem, tm, tn, (int)blockIdx.y);
}

// Grouped weight gradients: up to W4_MAX_GROUP independent TN products out_p[M_p, N_p] (+)= dy_p[T, M_p]^T x_p[T, N_p] over the SAME
// token count in ONE launch, one block per output tile of any of them.  The CLIP tower's four Linears of a layer have 48 + 16 + 64 +
// 64 = 192 tiles of 256^2 between them: launched one by one each needs split-K with fp32 partials to occupy the chip (0.13 of the MFMA
// peak); together they fill 192 of 256 CUs for the whole contraction, no partials, no reduce pass.
constexpr int W4_MAX_GROUP = 8;
struct W4Prob {
  const uint16_t* A;
  const uint16_t* B;
  void* C;
  int64_t lda, ldb, ldc;
  int M, N, epi, tiles_m, tiles_n, tile_end;  // tile_end: running total of tiles up to and including this problem
};
struct W4Group {
  W4Prob p[W4_MAX_GROUP];
  int n, K, gm;
};

template <int DT>
__global__ __launch_bounds__(256, 1) void gemm_w4_grouped(W4Group G) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // consecutive blocks go to consecutive XCDs: give each XCD one contiguous run of the concatenated tile list
  const int nwg = gridDim.x, v = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = v & 7, idx = v >> 3;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  int pi = 0;
#pragma unroll
  for (int i = 0; i < W4_MAX_GROUP - 1; ++i)
    if (i + 1 < G.n && tile >= G.p[i].tile_end) pi = i + 1;
  const W4Prob& P = G.p[pi];
  const int local = tile - (pi ? G.p[pi - 1].tile_end : 0);
  GemmArgs g{};
  g.A = P.A; g.B = P.B; g.C = P.C; g.lda = P.lda; g.ldb = P.ldb; g.ldc = P.ldc;
  g.M = P.M; g.N = P.N; g.K = G.K; g.epi = P.epi; g.tiles_m = P.tiles_m; g.tiles_n = P.tiles_n; g.vec_ok = 1; g.splits = 1; g.gm = G.gm;
  const int per_group = g.gm * g.tiles_n;
  const int group = local / per_group, first_m = group * g.gm;
  const int gsize = min(g.tiles_m - first_m, g.gm);
  const int in_g = local - group * per_group;
  w4_tile<DT, true, true, EK_STD>(g, smem, first_m + in_g % gsize, in_g / gsize, 0);
}

// ---- fp8 operands (e4m3 bytes, per-row fp32 scales): the fp8 training step's NT products on the 4-wave structure ---------------------
// Same tile, LDS image and copies as the 16-bit NT form with K counted in 2-byte units (a 128-byte LDS row = 128 k).  One
// v_mfma_scale_f32_16x16x128_f8f6f4 per accumulator tile and K-tile consumes BOTH 16-byte pieces of a fragment row (chunks kq and 4 + kq),
// so a K-tile is 64 MFMAs of 32 cycles instead of 128 of 16, and the fragment double-buffering is by operand instead of by k-step:
//   A fragments (8 x 8 registers) live in two sets that alternate per K-tile; B fragments are two half sets (j 0..3 / 4..7).
//   phase E: 32 MFMAs with B[0..3] | read B[4..7] of tile t, lgkmcnt(0) + barrier B1 (buffer t&1 free), 10 copies of tile t+2
//   phase O: 32 MFMAs with B[4..7] | vmcnt(10) + barrier B2 (tile t+1 landed), read A (other set) and B[0..3] of tile t+1, 6 copies
// (up to three single-issue instructions behind a 32-cycle MFMA).  Hardware block scales stay 1.0; the per-row scales of both operands
// are applied to the accumulators in the epilogue.  Weight block exponents are NOT handled here: the host sends only exponent-free
// operands (weight gradients always; forward / dgrad weights while the quantiser's flag says every exponent is zero).
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void mfma_f8_acc(f32x4_t& c, const i32x8& a, const i32x8& b, int scale) {
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+a"(c) : "v"(a), "v"(b), "v"(scale));
}

template <int DT, int EK>
__global__ __launch_bounds__(256, 1) void gemm_w4_f8(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int tm, tn;
  tile_of_block(g, tm, tn);
  const int m0 = tm * 256, n0 = tn * (EK == EK_SWIGLU ? 128 : 256);
  const int nk = g.K / BK;
  using std::integral_constant;

  int voffA[8], voffB[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = wave * 64 + j * 8 + (lane >> 3);
    const int c = ((lane & 7) ^ ((row >> 1) & 7)) * 16;
    voffA[j] = (int)((int64_t)min(m0 + row, g.M - 1) * g.lda * 2 + c);
    int src = n0 + row;
    if constexpr (EK == EK_SWIGLU) {  // part rows [64 q, 64 q + 64): q = 2 * (column half) + (0 gate | 1 up), as in the 16-bit kernel
      const int q = row >> 6;
      src = (q & 1) * g.sw_ff + n0 + (q >> 1) * 64 + (row & 63);
    }
    voffB[j] = (int)((int64_t)min(src, g.N - 1) * g.ldb * 2 + c);
  }
  auto make_rs = [](const void* p_) {
    const uint64_t a_ = (uint64_t)(uintptr_t)p_;
    return i32x4{__builtin_amdgcn_readfirstlane((int)(uint32_t)a_),
                 __builtin_amdgcn_readfirstlane((int)(uint32_t)((a_ >> 32) & 0xffffu)), -1, 0x00020000};
  };
  const i32x4 rsA = make_rs(g.A), rsB = make_rs(g.B);
  const unsigned lds0 = lds_addr_of(smem);
  const unsigned w_kc = lds0 + (unsigned)wave * 8192u;
  auto copy_m0 = [&](auto C_, auto BUF_) {
    constexpr int c = decltype(C_)::value, bu = decltype(BUF_)::value;
    constexpr int imm = bu * W4_UNIT + (c >> 3) * W4_PART + (c & 7) * 1024;
    const unsigned base_ = w_kc;
    asm volatile("s_add_u32 m0, %0, %1" ::"s"(base_), "n"(imm) : "scc");
  };
  auto copy_ld = [&](auto C_, int t) {
    constexpr int c = decltype(C_)::value;
    const int vo = c < 8 ? voffA[c & 7] : voffB[c & 7];
    const i32x4 rs = c < 8 ? rsA : rsB;
    const unsigned soff = (unsigned)t * (unsigned)(BK * 2);
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(vo), "s"(rs), "s"(soff) : "memory");
  };

  f32x4_t acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, kq = lane >> 4;
  const unsigned swz = (unsigned)((fr >> 1) & 7);
  unsigned a_ad[2][2], b_ad[2][2];  // [buffer][16-byte piece: chunk kq / 4 + kq]
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      a_ad[b][h] = lds0 + (unsigned)b * W4_UNIT + (unsigned)(wm * 128 + fr) * 128 + (((4 * h + kq) ^ swz) << 4);
      b_ad[b][h] = lds0 + (unsigned)b * W4_UNIT + W4_PART + (unsigned)(wn * 128 + fr) * 128 + (((4 * h + kq) ^ swz) << 4);
    }
  u32x4 fa[2][8][2];  // A fragments [set][row block i][piece]
  u32x4 fb[8][2];     // B fragments [column block j][piece]
  const int one = 127;  // E8M0 1.0 for both hardware block scales

  // read instruction r of the A fragments of buffer BUF into set P (r = 2 i + piece); of B fragments j0..j0+3 (r = 2 (j - j0) + piece)
  auto read_a = [&](auto BUF_, auto P_, auto R_) {
    constexpr int bu = decltype(BUF_)::value, pset = decltype(P_)::value, r = decltype(R_)::value;
    dsr128<(r >> 1) * 2048>(fa[pset][r >> 1][r & 1], a_ad[bu][r & 1]);
  };
  auto read_b = [&](auto BUF_, auto J0_, auto R_) {
    constexpr int bu = decltype(BUF_)::value, j0 = decltype(J0_)::value, r = decltype(R_)::value;
    dsr128<(j0 + (r >> 1)) * 2048>(fb[j0 + (r >> 1)][r & 1], b_ad[bu][r & 1]);
  };
  auto mfma8 = [&](auto P_, auto I_, auto J_) {
    constexpr int pset = decltype(P_)::value, i = decltype(I_)::value, j = decltype(J_)::value;
    const i32x8 a = {(int)fa[pset][i][0][0], (int)fa[pset][i][0][1], (int)fa[pset][i][0][2], (int)fa[pset][i][0][3],
                     (int)fa[pset][i][1][0], (int)fa[pset][i][1][1], (int)fa[pset][i][1][2], (int)fa[pset][i][1][3]};
    const i32x8 b = {(int)fb[j][0][0], (int)fb[j][0][1], (int)fb[j][0][2], (int)fb[j][0][3],
                     (int)fb[j][1][0], (int)fb[j][1][1], (int)fb[j][1][2], (int)fb[j][1][3]};
    // (operands swapped like the 16-bit kernels: the accumulator holds 4 consecutive n of one row m)
    mfma_f8_acc(acc[i][j], b, a, one);
  };
  using I0 = integral_constant<int, 0>;
  using I1 = integral_constant<int, 1>;
  using I4 = integral_constant<int, 4>;
  // K-tile t from buffer BUF with A set P (both = t & 1)
  auto tile = [&](auto BUF_, int t) {
    constexpr int bu = decltype(BUF_)::value;
    using NB = integral_constant<int, 1 - bu>;
    const int tl = min(t + 2, nk - 1);
    // phase E
    w4_for<32>([&](auto SL_) {
      constexpr int sl = decltype(SL_)::value;
      mfma8(BUF_, integral_constant<int, sl % 8>{}, integral_constant<int, sl / 8>{});
      if constexpr (sl < 8) read_b(BUF_, I4{}, SL_);
      if constexpr (sl == 10) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      if constexpr (sl >= 11 && sl <= 29 && ((sl - 11) & 1) == 0) {
        copy_m0(integral_constant<int, (sl - 11) / 2>{}, BUF_);
        copy_ld(integral_constant<int, (sl - 11) / 2>{}, tl);
      }
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W4_FENCE();
    // phase O
    w4_for<32>([&](auto SL_) {
      constexpr int sl = decltype(SL_)::value;
      mfma8(BUF_, integral_constant<int, sl % 8>{}, integral_constant<int, 4 + sl / 8>{});
      if constexpr (sl == 1) {
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      if constexpr (sl >= 2 && sl < 18) read_a(NB{}, NB{}, integral_constant<int, sl - 2>{});
      if constexpr (sl >= 18 && sl < 26) read_b(NB{}, I0{}, integral_constant<int, sl - 18>{});
      if constexpr (sl >= 3 && sl <= 23 && ((sl - 3) & 3) == 0) {
        copy_m0(integral_constant<int, 10 + (sl - 3) / 4>{}, BUF_);
        copy_ld(integral_constant<int, 10 + (sl - 3) / 4>{}, tl);
      }
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W4_FENCE();
  };

  // prologue: tiles 0 and 1 in flight, tile 0 landed, its A fragments (set 0) and B[0..3] in registers
  w4_for<16>([&](auto C_) { copy_m0(C_, I0{}); copy_ld(C_, 0); });
  w4_for<16>([&](auto C_) { copy_m0(C_, I1{}); copy_ld(C_, 1); });
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  W4_FENCE();
  w4_for<16>([&](auto R_) { read_a(I0{}, I0{}, R_); });
  w4_for<8>([&](auto R_) { read_b(I0{}, I0{}, R_); });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  W4_FENCE();
  for (int t = 0; t < nk; t += 2) {  // nk is even (host)
    tile(I0{}, t);
    tile(I1{}, t + 1);
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");

  // store phase shared with the 16-bit kernel: C[m, n] = sc_m[m] * sc_n[n] * acc through the epilogue kind EK (plain / residual / accumulate, fp32 store,
  // RoPE and SwiGLU on the scaled fp32 accumulators, SwiGLU backward with its gu pieces requested ahead)
  w4_store<DT, EK, true>(g, smem, acc, m0, n0, 0);
}

template <int DT, bool AKS, bool BKS, int EK, int MI = 8>
int launch_w4(const GemmArgs& g, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_w4<DT, AKS, BKS, EK, MI>, hipFuncAttributeMaxDynamicSharedMemorySize, MI == 8 ? W4_LDS : W4_LDS_HALF);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_w4<DT, AKS, BKS, EK, MI>), dim3(g.tiles_m * g.tiles_n, g.splits > 1 ? g.splits : 1), dim3(256), MI == 8 ? W4_LDS : W4_LDS_HALF, stream, g);
  MH_LAUNCH_CHECK();
}

}  // namespace

// The epilogue kind a problem needs (-1: not one gemm_w4 has).
static int w4_kind(const GemmArgs& g) {
  if (g.sw_mode > 2) return -1;  // (quick-GELU forms: 8-wave kernel only)
  if (g.sw_mode == 1) return EK_SWIGLU;
  if (g.sw_mode == 2) return EK_SWIGLU_BWD;
  if (g.rope_tab) return EK_ROPE;
  if (g.epi & MH_EPI_OUT_F32) {
    if (g.epi == MH_EPI_OUT_F32) return EK_F32;
    if (g.epi == (MH_EPI_OUT_F32 | MH_EPI_ACCUM) && g.splits == 1) return EK_F32ACC;
    return -1;
  }
  const int e = g.epi & ~MH_EPI_ACCUM;
  return (e == 0 || e == MH_EPI_RESIDUAL) ? EK_STD : -1;
}

// true when gemm_w4 can run this problem.  K-contiguous operands need whole K-tile pairs (K % 128 == 0); a product of two K-strided
// operands (weight gradients) takes any K.  16-bit outputs go through the staged store (16-byte row pieces), fp32 outputs straight
// from the accumulators.  (The policy - where it is FASTER - lives in gemm.hip.)
bool w4_can_run(const GemmArgs& g, int a_kstrided, int b_kstrided) {
  const int kind = w4_kind(g);
  if (kind < 0 || !g.vec_ok) return false;
  const bool f32 = kind == EK_F32 || kind == EK_F32ACC;
  if (!f32 && ((g.N % 8) || (g.ldc % 8) || ((((uintptr_t)g.C) & 15u) != 0))) return false;
  if (f32 && ((g.N % 4) || (g.ldc % 4) || ((((uintptr_t)g.C) & 15u) != 0) || (g.c_split % 4))) return false;
  if (g.splits > 1 && kind != EK_F32) return false;  // split-K writes fp32 partials
  if (!(a_kstrided && b_kstrided) && g.K % (2 * BK) != 0) return false;
  if (g.splits > 1) {  // even K-tile counts per split, no empty split
    const int nkt = (g.K + 2 * BK - 1) / (2 * BK) * 2, per = ((nkt / 2 + g.splits - 1) / g.splits) * 2;
    if ((g.splits - 1) * per >= nkt) return false;
  }
  if (kind == EK_ROPE && (a_kstrided || b_kstrided || g.rope_D != 128 || (g.rope_cols % 128) || g.rope_S <= 0)) return false;
  if (kind == EK_SWIGLU && (a_kstrided || b_kstrided || (g.sw_ff % 8) || (g.sw_ldo % 8) || ((((uintptr_t)g.sw_out) & 15u) != 0))) return false;
  if (kind == EK_SWIGLU_BWD && ((g.sw_ff % 8) || (g.sw_ldo % 8) || (g.sw_ldi % 8) || ((((uintptr_t)g.sw_out) | ((uintptr_t)g.sw_in)) & 15u) != 0 ||
                                g.N != g.sw_ff))
    return false;
  if (a_kstrided && (g.M % 8)) return false;
  if (b_kstrided && (g.N % 8)) return false;
  const int64_t lim = (1ll << 32) - (1 << 20);
  const int64_t spanA = a_kstrided ? (int64_t)(g.K + 2 * BK) * g.lda * 2 : (int64_t)g.M * g.lda * 2;
  const int64_t spanB = b_kstrided ? (int64_t)(g.K + 2 * BK) * g.ldb * 2 : (int64_t)g.N * g.ldb * 2;
  return spanA < lim && spanB < lim;
}

// fp8 form: both operands K-contiguous bytes (g.K, lda, ldb in 2-byte units), no weight block exponents, an even number of 128-byte K-tiles; epilogue
// kinds: plain / residual / accumulating 16-bit store, fp32 store (lm_head), RoPE, SwiGLU forward and backward (all NT products in the fp8 step)
bool w4_f8_can_run(const GemmArgs& g) {
  const int kind = w4_kind(g);
  if (kind < 0 || kind == EK_F32ACC || !g.vec_ok) return false;
  if (kind == EK_F32) {
    if ((g.N % 4) || (g.ldc % 4) || ((((uintptr_t)g.C) & 15u) != 0)) return false;
  } else if ((g.N % 8) || (g.ldc % 8) || ((((uintptr_t)g.C) & 15u) != 0)) {
    return false;
  }
  if (g.K % (2 * BK) != 0 || g.splits != 1 || g.sc_e) return false;
  if (kind == EK_ROPE && (g.rope_D != 128 || (g.rope_cols % 128) || g.rope_S <= 0)) return false;
  if (kind == EK_SWIGLU && ((g.sw_ff % 8) || (g.sw_ldo % 8) || ((((uintptr_t)g.sw_out) & 15u) != 0))) return false;
  if (kind == EK_SWIGLU_BWD && ((g.sw_ff % 8) || (g.sw_ldo % 8) || (g.sw_ldi % 8) || ((((uintptr_t)g.sw_out) | ((uintptr_t)g.sw_in)) & 15u) != 0 || g.N != g.sw_ff))
    return false;
  if ((((uintptr_t)g.sc_n) & 15u) != 0 || (g.N % 4)) return false;
  const int64_t lim = (1ll << 32) - (1 << 20);
  return (int64_t)g.M * g.lda * 2 < lim && (int64_t)g.N * g.ldb * 2 < lim;
}
bool w4_f8_is_fused(const GemmArgs& g) { const int k = w4_kind(g); return k == EK_ROPE || k == EK_SWIGLU || k == EK_SWIGLU_BWD || k == EK_F32; }
template <int DT, int EK>
static int launch_w4_f8(const GemmArgs& g, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_w4_f8<DT, EK>, hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_w4_f8<DT, EK>), dim3(g.tiles_m * g.tiles_n), dim3(256), W4_LDS, stream, g);
  MH_LAUNCH_CHECK();
}
int launch_gemm_w4_f8(const GemmArgs& g, int dt, hipStream_t stream) {
  const int kind = w4_kind(g);
#define F8_GO(EK_) return dt == MH_BF16 ? launch_w4_f8<MH_BF16, EK_>(g, stream) : launch_w4_f8<MH_F16, EK_>(g, stream)
  switch (kind) {
    case EK_STD: F8_GO(EK_STD);
    case EK_F32: F8_GO(EK_F32);
    case EK_ROPE: F8_GO(EK_ROPE);
    case EK_SWIGLU: F8_GO(EK_SWIGLU);
    case EK_SWIGLU_BWD: F8_GO(EK_SWIGLU_BWD);
    default: break;
  }
#undef F8_GO
  return MH_ERR_ARG;
}

// (instantiated: every layout for the plain kinds; the fused kinds in the layout their call sites have)
// half = 1: the 128-row block tile (g.tiles_m counts 128-row tiles); NT products only (w4_has_half)
int launch_gemm_w4(const GemmArgs& g, int dt, int a_kstrided, int b_kstrided, hipStream_t stream, int half) {
  const int kind = w4_kind(g);
  const int lay = (a_kstrided ? 2 : 0) | (b_kstrided ? 1 : 0);
  if (half) {
    if (lay != 0) return MH_ERR_ARG;
#define W4_HALF(EK_) return dt == MH_BF16 ? launch_w4<MH_BF16, false, false, EK_, 4>(g, stream) : launch_w4<MH_F16, false, false, EK_, 4>(g, stream)
    switch (kind) {
      case EK_STD: W4_HALF(EK_STD);
      case EK_F32: W4_HALF(EK_F32);
      case EK_F32ACC: W4_HALF(EK_F32ACC);
      case EK_ROPE: W4_HALF(EK_ROPE);
      case EK_SWIGLU: W4_HALF(EK_SWIGLU);
      default: return MH_ERR_ARG;
    }
#undef W4_HALF
  }
#define W4_GO(DT_, AKS_, BKS_, EK_) return launch_w4<DT_, AKS_, BKS_, EK_>(g, stream)
#define W4_LAYOUTS(DT_, EK_)                                                     \
  switch (lay) {                                                                 \
    case 0: W4_GO(DT_, false, false, EK_);                                       \
    case 1: W4_GO(DT_, false, true, EK_);                                        \
    case 2: W4_GO(DT_, true, false, EK_);                                        \
    default: W4_GO(DT_, true, true, EK_);                                        \
  }
  if (dt == MH_BF16) {
    switch (kind) {
      case EK_STD: W4_LAYOUTS(MH_BF16, EK_STD)
      case EK_F32: if (lay == 0) W4_GO(MH_BF16, false, false, EK_F32); if (lay == 3) W4_GO(MH_BF16, true, true, EK_F32); break;
      case EK_F32ACC: if (lay == 0) W4_GO(MH_BF16, false, false, EK_F32ACC); break;
      case EK_ROPE: if (lay == 0) W4_GO(MH_BF16, false, false, EK_ROPE); break;
      case EK_SWIGLU: if (lay == 0) W4_GO(MH_BF16, false, false, EK_SWIGLU); break;
      case EK_SWIGLU_BWD: if (lay == 1) W4_GO(MH_BF16, false, true, EK_SWIGLU_BWD); break;
      default: break;
    }
  } else {
    switch (kind) {
      case EK_STD: W4_LAYOUTS(MH_F16, EK_STD)
      case EK_F32: if (lay == 0) W4_GO(MH_F16, false, false, EK_F32); if (lay == 3) W4_GO(MH_F16, true, true, EK_F32); break;
      case EK_F32ACC: if (lay == 0) W4_GO(MH_F16, false, false, EK_F32ACC); break;
      case EK_ROPE: if (lay == 0) W4_GO(MH_F16, false, false, EK_ROPE); break;
      case EK_SWIGLU: if (lay == 0) W4_GO(MH_F16, false, false, EK_SWIGLU); break;
      case EK_SWIGLU_BWD: if (lay == 1) W4_GO(MH_F16, false, true, EK_SWIGLU_BWD); break;
      default: break;
    }
  }
#undef W4_LAYOUTS
#undef W4_GO
  return MH_ERR_ARG;
}
// layouts each fused / fp32 kind is instantiated for (w4_can_run is necessary, this is the second condition)
bool w4_has_kernel(const GemmArgs& g, int a_kstrided, int b_kstrided) {
  const int kind = w4_kind(g), lay = (a_kstrided ? 2 : 0) | (b_kstrided ? 1 : 0);
  switch (kind) {
    case EK_STD: return true;
    case EK_F32: return lay == 0 || lay == 3;
    case EK_F32ACC: case EK_ROPE: case EK_SWIGLU: return lay == 0;
    case EK_SWIGLU_BWD: return lay == 1;
    default: return false;
  }
}

// the 128-row block tile exists for the NT products (forward: a short prefill's projections, the fp32 logits)
bool w4_has_half(const GemmArgs& g, int a_kstrided, int b_kstrided) {
  const int kind = w4_kind(g);
  return !a_kstrided && !b_kstrided && (kind == EK_STD || kind == EK_F32 || kind == EK_F32ACC || kind == EK_ROPE || kind == EK_SWIGLU);
}

// grouped weight gradients (see gemm_w4_grouped): n <= W4_MAX_GROUP problems over the same K
int launch_gemm_w4_grouped(const GemmArgs* probs, int n, int dt, int gm, hipStream_t stream) {
  if (n < 1 || n > W4_MAX_GROUP) return MH_ERR_ARG;
  W4Group G{};
  G.n = n; G.K = probs[0].K; G.gm = gm;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    const GemmArgs& g = probs[i];
    if (g.K != G.K || !w4_can_run(g, 1, 1) || w4_kind(g) != EK_STD) return MH_ERR_ARG;
    W4Prob& P = G.p[i];
    P.A = g.A; P.B = g.B; P.C = g.C; P.lda = g.lda; P.ldb = g.ldb; P.ldc = g.ldc; P.M = g.M; P.N = g.N; P.epi = g.epi;
    P.tiles_m = g.tiles_m; P.tiles_n = g.tiles_n;
    total += g.tiles_m * g.tiles_n;
    P.tile_end = total;
  }
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_w4_grouped<MH_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS);
    hipFuncSetAttribute((const void*)gemm_w4_grouped<MH_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS);
    attr_set = true;
  }
  if (dt == MH_BF16) hipLaunchKernelGGL((gemm_w4_grouped<MH_BF16>), dim3(total), dim3(256), W4_LDS, stream, G);
  else hipLaunchKernelGGL((gemm_w4_grouped<MH_F16>), dim3(total), dim3(256), W4_LDS, stream, G);
  MH_LAUNCH_CHECK();
}

}  // namespace mhgemm

#ifdef MH_W4_TIMING
extern "C" int mh_w4_timing_buffer(void* buf) {  // 8 x uint64 per block of the next gemm_w4 launches (nullptr: off)
  unsigned long long* p = (unsigned long long*)buf;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(mhgemm::g_w4_dbg), &p, sizeof(p));
}
#endif
