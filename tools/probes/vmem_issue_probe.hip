// What a tile-copy instruction costs a wave that is otherwise issuing MFMAs back to back (profiles/r05_vmem_issue_probe.txt; HISTORY §5 / §7 (0)).
// Every wave loops over {PER independent 16x16x32 MFMAs (inline asm, AccVGPR accumulators); one vector-memory instruction of form MODE}; one block per CU
// (140 KiB of LDS), 256 threads = one wave per SIMD (WAVES = 4) or 512 = two per SIMD (WAVES = 8).  The data is L2-resident (each CU walks its own window: 64 KiB = L2-resident footprint, 1 MiB = Infinity Cache / HBM).
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/vmem_issue_probe.hip -o tools/probes/vmem_issue_probe.so ; python tools/probes/vmem_issue_probe.py
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// MODE 0 none | 1 m0 write + buffer_load_dwordx4 offen lds | 2 buffer_load_dwordx4 offen -> registers | 3 dwordx2 -> registers | 4 dword -> registers
//      5 global_load_lds_dwordx4 (saddr + voffset) | 6 = 1 with 32 active lanes | 7 = 2 with a wave-uniform address (off, no VGPR) | 8 = 1, two per slot and
//      half as many slots (bursts of two) | 9 ds_read_b128 instead (LDS fragment read, for scale)
// GEMM-like surroundings (G): 0 none | 1 two ds_read_b128 per slot + lgkmcnt(0) every 8 slots | 2 = 1 + s_barrier every 8 slots | 3 = 2 + vmcnt(8) in front of each barrier
template <int MODE, int PER, int G = 0>
__global__ __launch_bounds__(512, 1) void probe_k(const char* __restrict__ src, float* __restrict__ sink, int iters, unsigned window) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x4 acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 a = {0x3f803f80u + lane, 0x3f003f80u, 0x3e803f00u, 0x3f803e80u}, b = {0x3f803f00u, 0x3f803f80u + tid, 0x3f003f00u, 0x3e803e80u};
  const uint64_t base = (uint64_t)(uintptr_t)src + (uint64_t)blockIdx.x * window;  // window: bytes a CU walks (64 KiB: the chip's footprint fits the L2s; 1 MiB: it does not)
  const i32x4 rs = {__builtin_amdgcn_readfirstlane((int)(uint32_t)base), __builtin_amdgcn_readfirstlane((int)(uint32_t)((base >> 32) & 0xffffu)), -1, 0x00020000};
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (unsigned)wave * 16384u;
  int vo = lane * 16 + wave * 1024;  // 1 KiB contiguous per instruction
  const unsigned lds_rd = lds0 + (unsigned)lane * 16u;
  unsigned soff = 0;
  u32x4 sink4 = {0, 0, 0, 0};
  u32x4 fr[8];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {  // 16 "copy slots" per iteration
#pragma unroll
      for (int m = 0; m < PER; ++m) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[(c * PER + m) & 15]) : "v"(a), "v"(b));
      }
      if constexpr (G >= 1) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(fr[(2 * c) & 7]) : "v"(lds_rd) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(fr[(2 * c + 1) & 7]) : "v"(lds_rd) : "memory");
      }
      const unsigned so = soff + (unsigned)c * 4096u;
      if constexpr (MODE == 1) {
        asm volatile("s_add_u32 m0, %0, %1" ::"s"(lds0), "n"(0) : "scc");
        asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(vo), "s"(rs), "s"(so) : "memory");
      } else if constexpr (MODE == 2) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(sink4) : "v"(vo), "s"(rs), "s"(so) : "memory");
      } else if constexpr (MODE == 3) {
        u32x2 s2;
        asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(s2) : "v"(vo), "s"(rs), "s"(so) : "memory");
      } else if constexpr (MODE == 4) {
        unsigned s1;
        asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(s1) : "v"(vo), "s"(rs), "s"(so) : "memory");
      } else if constexpr (MODE == 5) {
        const uint64_t sb = base + so;
        asm volatile("s_add_u32 m0, %0, %1" ::"s"(lds0), "n"(0) : "scc");
        asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(sb) : "memory");
      } else if constexpr (MODE == 6) {
        asm volatile("s_add_u32 m0, %0, %1" ::"s"(lds0), "n"(0) : "scc");
        asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, 0xffffffff\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_mov_b64 exec, s[20:21]" ::"v"(vo), "s"(rs), "s"(so)
                     : "memory", "s20", "s21");
      } else if constexpr (MODE == 7) {
        asm volatile("buffer_load_dwordx4 %0, off, %1, %2" : "=v"(sink4) : "s"(rs), "s"(so) : "memory");
      } else if constexpr (MODE == 8) {
        if (c & 1) {
          asm volatile("s_add_u32 m0, %0, %1" ::"s"(lds0), "n"(0) : "scc");
          asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(vo), "s"(rs), "s"(so) : "memory");
          asm volatile("s_add_u32 m0, %0, %1" ::"s"(lds0), "n"(1024) : "scc");
          asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(vo), "s"(rs), "s"(so + 2048u) : "memory");
        }
      } else if constexpr (MODE == 9) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(sink4) : "v"(lds_rd) : "memory");
      } else if constexpr (MODE == 10) {  // LDS-DMA with a wave-uniform address (no VGPR operand at all)
        asm volatile("s_add_u32 m0, %0, %1" ::"s"(lds0), "n"(0) : "scc");
        asm volatile("buffer_load_dwordx4 off, %0, %1 lds" ::"s"(rs), "s"(so) : "memory");
      } else if constexpr (MODE == 11) {  // the copy as two 32-lane halves
        asm volatile("s_add_u32 m0, %0, %1" ::"s"(lds0), "n"(0) : "scc");
        asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, 0xffffffff\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_mov_b64 exec, s[20:21]" ::"v"(vo), "s"(rs), "s"(so)
                     : "memory", "s20", "s21");
        asm volatile("s_add_u32 m0, %0, %1" ::"s"(lds0), "n"(512) : "scc");
        asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, 0xffffffff\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_mov_b64 exec, s[20:21]" ::"v"(vo), "s"(rs), "s"(so + 512u)
                     : "memory", "s20", "s21");
      }
      if constexpr (G >= 1) {
        if ((c & 7) == 7) {
          if constexpr (G >= 3 && MODE != 0 && MODE != 9) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if constexpr (G >= 2) __builtin_amdgcn_s_barrier();
        }
      }
      if constexpr (MODE != 0 && MODE != 9) {
        if (c == 15) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // one iteration's requests in flight
      } else if constexpr (MODE == 9) {
        if (c == 15) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    soff = (soff + 65536u) & (window - 1u);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += acc[j][0] + acc[j][3];
  s += (float)(sink4[0] & 1u);
  if constexpr (G >= 1) s += (float)(fr[0][0] & 1u) + (float)(fr[7][3] & 1u);
  if (s == 12345.678f) sink[blockIdx.x * blockDim.x + tid] = s;
}

template <int MODE, int PER, int G = 0>
static float run(const char* src, float* sink, int iters, int waves, int blocks, int reps, unsigned window) {
  hipFuncSetAttribute((const void*)probe_k<MODE, PER, G>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe_k<MODE, PER, G>), dim3(blocks), dim3(64 * waves), 140 * 1024, 0, src, sink, iters, window);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe_k<MODE, PER, G>), dim3(blocks), dim3(64 * waves), 140 * 1024, 0, src, sink, iters, window);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms / reps;
}

// ns per copy slot (= PER MFMAs + one instruction of the form) and wave; src >= 256 MiB, sink >= blocks * 512 floats
extern "C" float vmem_issue_probe(int mode, int per, const void* src, void* sink, int iters, int waves, int blocks, int reps, unsigned window) {
  const char* s = (const char*)src;
  float* k = (float*)sink;
  float ms = -1.f;
#define GO(M, P) if (mode == M && per == P) ms = run<M, P>(s, k, iters, waves, blocks, reps, window);
#define GG(M, G_) if (mode == M + 100 * G_ && per == 8) ms = run<M, 8, G_>(s, k, iters, waves, blocks, reps, window);
  GG(0, 1) GG(1, 1) GG(2, 1) GG(0, 2) GG(1, 2) GG(2, 2) GG(0, 3) GG(1, 3) GG(2, 3) GG(4, 1) GG(6, 1) GG(7, 1) GG(8, 1) GG(10, 1) GG(11, 1)
#undef GG
#define ALLP(M) GO(M, 4) GO(M, 8) GO(M, 16)
  ALLP(0) ALLP(1) ALLP(2) ALLP(3) ALLP(4) ALLP(5) ALLP(6) ALLP(7) ALLP(8) ALLP(9)
#undef ALLP
#undef GO
  if (ms < 0.f) return -1.f;
  return ms * 1e6f / ((float)iters * 16.f);
}
