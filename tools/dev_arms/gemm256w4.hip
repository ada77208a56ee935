// gemm_nt_w4: 256x256-tile bf16/f16 MFMA GEMM, FOUR waves per block, one 128x128 output quadrant per wave
// (gfx950).  K-contiguous operands only (the forward "NT" case); see gemm256.hip for the K-strided forms.
// STATUS: A/B arm (mh_gemm_force_kernel(4)), not dispatched by default - measured on the cfg-3 shapes it ties the
// 8-wave kernel on the big plain GEMMs (1.39-1.45 PFLOP/s) and loses where the epilogue reads a residual/bias or K is
// short (half as many waves share the same epilogue work); kept with its timing probes (VAR) as the record of what
// was learnt about one-wave-per-SIMD scheduling on gfx950 (HISTORY.md §3, GEMM rows).
//
// Why a second 256-tile kernel: with 8 waves (gemm256.hip) every wave owns 128x64 outputs, so a K-step of 32
// costs 12 fragment reads per 32 MFMAs and the block needs 8 barriers per K-tile.  Here a wave owns 128x128
// outputs = 64 accumulator tiles = 256 fp32 registers per lane, which only fit because gfx950's register file
// is 512 per lane at one wave per SIMD (the accumulators live in AGPRs, the fragments in VGPRs): a K-step of 32
// costs 16 fragment reads per 64 MFMAs (LDS traffic per flop halved), there is ONE barrier per K-step, and the
// MFMA stream of a wave is 64 independent instructions long, so a single wave per SIMD keeps the matrix core
// busy while its own LDS reads and global->LDS copies for later K-steps are in flight.
//
// LDS = two K-tile buffers of 64 KiB: A part [256 rows][128 B] + B part [256 rows][128 B] (full 128-byte lines per
// row: a first version staged 64-byte half rows per K-step, which doubled the number of L1->L2 read requests -
// TCP_TCC_READ_REQ 136 M vs 68 M - and capped the kernel at 1.1 PFLOP/s).  Fragments are double-buffered per
// K-step of 32 (register set 0 = k 0..31, set 1 = k 32..63).  Per K-tile t:
//   phase E: 64 MFMAs on set 0 | read set 1 (tile t, k 32..63) from buffer t&1
//   phase O: 64 MFMAs on set 1 | s_waitcnt vmcnt(0) + s_barrier (tile t+1 has landed - its copies were issued a
//            whole K-tile ago - and every wave has all of tile t in registers, so buffer t&1 is free), then issue
//            tile t+2's 16 copies per wave into buffer t&1 and read set 0 of tile t+1 from buffer (t+1)&1.
// One barrier per K-tile; a copy has two phases (128 MFMAs, > 1 us) to land.
//
// Row swizzle as in gemm.hip: 16-byte chunk c of row r is stored at c ^ ((r >> 1) & 7) (conflict-free ds_read_b128
// fragment reads); the copies are global_load_lds (lane-linear LDS image), so it is applied to the global source.
#include <type_traits>

#include "gemm_common.h"

namespace mhgemm {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// MFMA with the accumulator pinned to AGPRs ("+a"): hipcc's allocator otherwise shuttles part of the 256
// accumulators between VGPRs and AGPRs inside the loop.  Volatile: the issue order below IS the schedule.
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

template <int OFF>
__device__ __forceinline__ void dsr(u32x4& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
#define W4_FENCE() __builtin_amdgcn_sched_barrier(0)
template <int N, typename F>
__device__ __forceinline__ void w4_static_for(F&& f) {
  if constexpr (N > 0) {
    w4_static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

template <int DT, int VAR>
__global__ __launch_bounds__(256, 1) void gemm_nt_w4(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int tm, tn;
  tile_of_block(g, tm, tn);
  const int m0 = tm * 256, n0 = tn * 256;
  const int nk = g.K / BK;

  // Copies are raw buffer loads to LDS: per-lane byte offsets voff (row clamped to the operand, swizzled chunk) are
  // loop-invariant VGPRs and the K advance is the scalar soffset, so a copy costs NO VALU instruction in the loop
  // (a global_load_lds needs a 64-bit VALU address add per copy; that VALU->VMEM dependency stalled the in-order
  // MFMA stream whenever the address pipe was busy, and held this kernel at 1.15 PFLOP/s).
  // wave-load j (0..7) of this wave covers part rows 64*wave + 8j .. +7; lane i -> row + (i>>3), physical chunk
  // i&7 = logical chunk (i&7) ^ ((row>>1)&7)
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, 0x7fffffff, 0x00020000);
  int voffA[8], voffB[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = wave * 64 + j * 8 + (lane >> 3);
    const int c = ((lane & 7) ^ ((row >> 1) & 7)) * 16;
    voffA[j] = (int)((int64_t)min(m0 + row, g.M - 1) * g.lda * 2 + c);
    voffB[j] = (int)((int64_t)min(n0 + row, g.N - 1) * g.ldb * 2 + c);
  }
  // copy j (0..7: A rows, 8..15: B rows) of K-tile t -> buffer t&1 (prologue form: compiler-scheduled)
  auto issue1 = [&](auto J, int t) {
    constexpr int j = decltype(J)::value;
    char* dst = smem + (t & 1) * W4_UNIT + (j >> 3) * W4_PART + wave * 8192 + (j & 7) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(j < 8 ? rsA : rsB, (__attribute__((address_space(3))) void*)dst, 16,
                                             j < 8 ? voffA[j & 7] : voffB[j & 7], t * (BK * 2), 0, 0);
  };
  auto issue = [&](int t) { w4_static_for<16>([&](auto J) { issue1(J, t); }); };
  // loop form: the M0 write and the copy are separate single instructions, each placed behind its own MFMA
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  auto make_rs = [](const void* p_) {
    const uint64_t a_ = (uint64_t)(uintptr_t)p_;
    return i32x4{__builtin_amdgcn_readfirstlane((int)(uint32_t)a_),
                 __builtin_amdgcn_readfirstlane((int)(uint32_t)((a_ >> 32) & 0xffffu)), 0x7fffffff, 0x00020000};
  };
  const i32x4 rsAv = make_rs(g.A), rsBv = make_rs(g.B);
  const unsigned lds_w = lds_addr_of(smem) + (unsigned)wave * 8192;
  auto copy_m0 = [&](auto J, unsigned bufbase) {
    constexpr int j = decltype(J)::value;
    if constexpr (VAR != 6 && VAR != 7) asm volatile("s_add_u32 m0, %0, %1" ::"s"(bufbase), "n"((j >> 3) * W4_PART + (j & 7) * 1024) : "scc");
  };
  auto copy_ld = [&](auto J, int soff) {
    constexpr int j = decltype(J)::value;
    const int vo = j < 8 ? voffA[j & 7] : voffB[j & 7];
    const i32x4 rs = j < 8 ? rsAv : rsBv;
    if constexpr (VAR == 7) {  // probe: plain load to VGPRs (no LDS-DMA)
      u32x4 tmp;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(tmp) : "v"(vo), "s"(rs), "s"(soff) : "memory");
    } else {
      asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(vo), "s"(rs), "s"(soff) : "memory");
    }
  };

  f32x4_t acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, kq = lane >> 4;
  const unsigned lds0 = lds_addr_of(smem);
  const unsigned swz = (unsigned)((fr >> 1) & 7);
  // per k-step (register set) lane addresses inside buffer 0
  const unsigned a_lane[2] = {lds0 + (unsigned)(wm * 128 + fr) * 128 + (((0 + kq) ^ swz) << 4),
                              lds0 + (unsigned)(wm * 128 + fr) * 128 + (((4 + kq) ^ swz) << 4)};
  const unsigned b_lane[2] = {lds0 + W4_PART + (unsigned)(wn * 128 + fr) * 128 + (((0 + kq) ^ swz) << 4),
                              lds0 + W4_PART + (unsigned)(wn * 128 + fr) * 128 + (((4 + kq) ^ swz) << 4)};

  u32x4 af[2][8], bf[2][8];

  // fragment read q (0..7: A row tiles, 8..15: B row tiles) of K-tile buffer base `ub`, k-step SET -> register set SET
  auto read1 = [&](auto SET, auto Q, unsigned ub) {
    constexpr int s = decltype(SET)::value, q = decltype(Q)::value;
    if constexpr (q < 8) dsr<(q & 7) * 2048>(af[s][q & 7], a_lane[s] + ub);
    else dsr<(q & 7) * 2048>(bf[s][q & 7], b_lane[s] + ub);
  };
  using std::integral_constant;
  using S0 = integral_constant<int, 0>;
  using S1 = integral_constant<int, 1>;
  // MFMA slot sl (0..63): accumulator (sl/8, sl%8).  The matrix core takes an independent 16x16x32 MFMA every 16
  // cycles and a wave issues in order, so with ONE wave per SIMD at most one other instruction may sit between two
  // MFMAs: every fragment read, M0 write and copy below is placed behind its own MFMA.
  auto mfma_slot = [&](auto SET, auto SL) {
    constexpr int s = decltype(SET)::value, sl = decltype(SL)::value;
    mfma_acc<DT>(acc[sl % 8][sl / 8], bf[s][sl / 8], af[s][sl % 8]);  // srcA (B fragment) fixed for 8 consecutive MFMAs
  };
  // copy schedule of K-tile t+2 (-> buffer t&1), one M0 write + one copy per 6 MFMA slots so that the address
  // pipe sees a steady ~40 B/clk instead of a 128 B/clk burst: copies 0..6 at phase-E slots 24+6c, copies 7..15 at
  // phase-O slots 6+6(c-7); every copy has >= 75 slots (> 1200 cycles) before barrier B2 of the next tile.
  auto phase_e = [&](int t, auto LOADS) {  // MFMAs on set 0 (tile t, k 0..31); fetch set 1 of tile t
    constexpr bool loads = decltype(LOADS)::value && VAR != 1;
    const unsigned ub = (unsigned)(t & 1) * W4_UNIT;
    const unsigned m0base = lds_w + ub;
    const int soff = (t + 2) * (BK * 2);
    w4_static_for<64>([&](auto SL) {
      constexpr int sl = decltype(SL)::value;
      mfma_slot(S0{}, SL);
      if constexpr (sl < 16 && VAR != 2) read1(S1{}, integral_constant<int, sl>{}, ub);
      if constexpr (loads && sl == 23) {  // B1: every wave has all of tile t in registers -> buffer t&1 is free
        if constexpr (VAR != 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (VAR != 3) __builtin_amdgcn_s_barrier();
      }
      if constexpr (loads && sl >= 24 && sl < 66 && (sl - 24) % 6 < 2) {
        if constexpr ((sl - 24) % 6 == 0) copy_m0(integral_constant<int, (sl - 24) / 6>{}, m0base);
        else copy_ld(integral_constant<int, (sl - 24) / 6>{}, soff);
      }
    });
    if constexpr (VAR != 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W4_FENCE();
  };
  auto phase_o = [&](int t, auto LOADS) {  // MFMAs on set 1 (tile t, k 32..63); fetch set 0 of tile t+1
    constexpr bool loads = decltype(LOADS)::value && VAR != 1;
    const unsigned ub_next = (unsigned)((t + 1) & 1) * W4_UNIT;
    const unsigned m0base = lds_w + (unsigned)(t & 1) * W4_UNIT;
    const int soff = (t + 2) * (BK * 2);
    w4_static_for<64>([&](auto SL) {
      constexpr int sl = decltype(SL)::value;
      mfma_slot(S1{}, SL);
      if constexpr (sl == 3) {  // B2: tile t+1 has landed for every wave (the 7 newest copies belong to tile t+2)
        if constexpr (VAR != 1 && VAR != 4 && VAR != 5) {
          if constexpr (loads) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (VAR != 3) __builtin_amdgcn_s_barrier();
      }
      constexpr bool copy_slot = loads && sl >= 6 && sl < 60 && (sl % 6) < 2;
      if constexpr (copy_slot) {
        if constexpr (sl % 6 == 0) copy_m0(integral_constant<int, 7 + (sl - 6) / 6>{}, m0base);
        else copy_ld(integral_constant<int, 7 + (sl - 6) / 6>{}, soff);
      }
      // fragment reads take the other slots from 4 on: read index = number of non-copy slots in [4, sl)
      constexpr int nread = loads ? (sl - 4) - (sl >= 6 ? 2 * ((sl - 6) / 6) + ((sl - 6) % 6 >= 1 ? 1 : 0) + ((sl - 6) % 6 >= 2 ? 1 : 0) - ((sl - 6) % 6 >= 2 ? 0 : 0) : 0) : sl - 4;
      if constexpr (!copy_slot && sl >= 4 && nread >= 0 && nread < 16 && VAR != 2)
        read1(S0{}, integral_constant<int, nread>{}, ub_next);  // (past the last tile the reads fetch dead data, unused)
    });
    if constexpr (VAR != 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W4_FENCE();
  };

  // prologue: tiles 0 and 1 in flight, tile 0 landed, its k-step 0 fragments in registers
  issue(0);
  if (nk > 1) issue(1);
  if (nk > 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  W4_FENCE();
  w4_static_for<16>([&](auto Q) { read1(S0{}, Q, 0u); });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  W4_FENCE();

  for (int t = 0; t < nk - 2; ++t) {
    phase_e(t, std::true_type{});
    phase_o(t, std::true_type{});
  }
  for (int t = max(nk - 2, 0); t < nk; ++t) {  // the last two tiles have nothing left to prefetch
    phase_e(t, std::false_type{});
    phase_o(t, std::false_type{});
  }
  // the s_nops cover the MFMA -> accumulator-read hazard that the compiler cannot see through the inline-asm MFMAs
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");

  if constexpr (VAR == 8) {  // probe: no epilogue stores (keeps the accumulators alive through one store)
    if (g.M == 12345) epi_store4<DT>(g, m0, n0, acc[0][0][0] + acc[7][7][3] + acc[3][4][1], 0.f, 0.f, 0.f);
    return;
  }
  // Staged epilogue (16-bit C, vectorisable layout): the accumulator layout gives a lane 4 consecutive n of one
  // row, i.e. 32-byte pieces of 16 different rows per store instruction - measured at ~19 us per 256x256 tile, a
  // fifth of a K=4096 tile's run time.  Instead every wave packs its 128x128 quadrant into its own LDS slice
  // ([128 rows][272 B]: 256 B of data + 16 B pad, conflict-free for the 8-byte writes and the 16-byte reads) and
  // writes it out as 16 bytes per lane = 256 contiguous bytes per row, 4 rows per instruction.
  if (epi_can_stage(g)) {
    __syncthreads();  // every wave is done with the operand tiles in LDS
    char* stage = smem + wave * W4_CSTAGE;
    const unsigned st_w = lds_addr_of(stage) + (unsigned)(lane & 15) * 272 + (unsigned)(lane >> 4) * 8;
    auto fill = [&](auto EPI_) {
      constexpr int EPI = decltype(EPI_)::value;
      w4_static_for<64>([&](auto T) {
        constexpr int t = decltype(T)::value, i = t / 8, j = t % 8;
        const int m = min(m0 + wm * 128 + i * 16 + (lane & 15), g.M - 1);
        const int n = min(n0 + wn * 128 + j * 16 + 4 * (lane >> 4), g.N - 4);
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        epi_xform4<DT, EPI>(g, m, n, v);
        const uint2 pk = make_uint2(pack2<DT>(v[0], v[1]), pack2<DT>(v[2], v[3]));
        const unsigned sw_ = st_w;  // (local copy: clang rejects captured variables as asm operands in nested generic lambdas)
        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(sw_), "v"(pk), "n"(i * 16 * 272 + j * 32) : "memory");
      });
    };
    switch (g.epi) {
      case 0: fill(integral_constant<int, 0>{}); break;
      case MH_EPI_RESIDUAL: fill(integral_constant<int, MH_EPI_RESIDUAL>{}); break;
      case MH_EPI_BIAS: fill(integral_constant<int, MH_EPI_BIAS>{}); break;
      case MH_EPI_BIAS | MH_EPI_QUICK_GELU: fill(integral_constant<int, MH_EPI_BIAS | MH_EPI_QUICK_GELU>{}); break;
      case MH_EPI_BIAS | MH_EPI_RESIDUAL: fill(integral_constant<int, MH_EPI_BIAS | MH_EPI_RESIDUAL>{}); break;
      default: break;  // excluded by epi_can_stage
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned st_r = lds_addr_of(stage) + (unsigned)(lane >> 4) * 272 + (unsigned)(lane & 15) * 16;
    const int mrow = m0 + wm * 128 + (lane >> 4);
    const int ncol = n0 + wn * 128 + (lane & 15) * 8;
    uint16_t* cp = (uint16_t*)g.C + (int64_t)mrow * g.ldc + ncol;
    const bool n_ok = ncol < g.N;
#pragma unroll
    for (int half = 0; half < 4; ++half) {
      u32x4 rv[8];
      w4_static_for<8>([&](auto R) { constexpr int r = decltype(R)::value; dsr<r * 4 * 272>(rv[r], st_r + (unsigned)half * 32 * 272); });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      W4_FENCE();
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int row = half * 32 + r * 4;
        if (n_ok && mrow + row < g.M) *(u32x4*)(cp + (int64_t)row * g.ldc) = rv[r];
      }
    }
    return;
  }
  // (no generic fallback here: the host routes only staged-epilogue cases to this kernel - 64 inlined generic stores per
  // thread took 12 minutes to compile and spilled)
}

}  // namespace

template <int DT, int VAR>
int launch_w4(const GemmArgs& g, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_nt_w4<DT, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_nt_w4<DT, VAR>), dim3(g.tiles_m * g.tiles_n), dim3(256), W4_LDS, stream, g);
  MH_LAUNCH_CHECK();
}

// var: 0 = the kernel; 1..8 = timing probes with wrong results (compiled only with -DMH_W4_PROBES: each is a full kernel)
int launch_gemm_nt_w4(const GemmArgs& g, int dt, hipStream_t stream, int var) {
#ifdef MH_W4_PROBES
  if (var == 1) return launch_w4<MH_BF16, 1>(g, stream);
  if (var == 2) return launch_w4<MH_BF16, 2>(g, stream);
  if (var == 3) return launch_w4<MH_BF16, 3>(g, stream);
  if (var == 4) return launch_w4<MH_BF16, 4>(g, stream);
  if (var == 5) return launch_w4<MH_BF16, 5>(g, stream);
  if (var == 6) return launch_w4<MH_BF16, 6>(g, stream);
  if (var == 7) return launch_w4<MH_BF16, 7>(g, stream);
  if (var == 8) return launch_w4<MH_BF16, 8>(g, stream);
#else
  if (var != 0) return MH_ERR_ARG;
#endif
  return dt == MH_BF16 ? launch_w4<MH_BF16, 0>(g, stream) : launch_w4<MH_F16, 0>(g, stream);
}

}  // namespace mhgemm
