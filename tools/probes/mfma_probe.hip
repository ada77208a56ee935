// Sustained bf16 MFMA rate of one MI355X with NO memory traffic: every wave loops over 16 independent 16x16x32 accumulate chains on
// constant operands.  What the matrix pipes deliver once the chip has settled under its power cap - the ceiling any GEMM main loop is under.
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/mfma_probe.hip -o tools/probes/mfma_probe.so ; python tools/probes/mfma_probe.py
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;

template <int SHAPE, bool RND, int ORDER = 0>
__global__ __launch_bounds__(256) void mfma_k(float* __restrict__ sink, int iters, float seed) {
  // 8 different operand pairs, used in rotation: RND = pseudo-random bit patterns (operand buses toggle as in a real GEMM), else near-constant
  bf16x8_t av[8], bv[8];
  unsigned st = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u) ^ 0x9e3779b9u;
  for (int q = 0; q < 8; ++q)
    for (int i = 0; i < 8; ++i) {
      st = st * 1664525u + 1013904223u;
      const float ra = RND ? ((int)(st >> 9) - (1 << 22)) * (1.0f / (1 << 20)) : seed + threadIdx.x * 1e-3f + i;
      st = st * 1664525u + 1013904223u;
      const float rb = RND ? ((int)(st >> 9) - (1 << 22)) * (1.0f / (1 << 20)) : seed - i * 0.5f;
      av[q][i] = (__bf16)ra; bv[q][i] = (__bf16)rb;
    }
  float s = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4_t acc[16];
    for (int j = 0; j < 16; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        // ORDER 0: both operands change on every MFMA; 1: the first operand is held for 4 consecutive MFMAs; 2: the second for 4;
        // 3: the GEMM's quad order (first alternates between two values, second held for two)
        constexpr int dummy = 0; (void)dummy;
        const int ia = ORDER == 1 ? (j >> 2) : ORDER == 3 ? (j & 1) + 2 * (j >> 3) : ORDER == 2 ? j & 7 : j & 7;
        const int ib = ORDER == 2 ? (j >> 2) : ORDER == 3 ? (j >> 1) & 3 : ORDER == 1 ? j & 7 : (j + 3) & 7;
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[ia & 7], bv[ib & 7], acc[j], 0, 0, 0);
      }
    }
    for (int j = 0; j < 16; ++j) s += acc[j][0] + acc[j][3];
  } else {
    f32x16_t acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[j], bv[(j + 3) & 7], acc[j & 3], 0, 0, 0);
    }
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][15];
  }
  if (s == 12345.678f) sink[blockIdx.x] = s;
}

// flops per launch = blocks * 4 waves * iters * (16 * 16x16x32x2  |  8 * 32x32x16x2)
extern "C" int mfma_run(void* sink, int blocks, int iters, int shape, int rnd, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (shape == 161) hipLaunchKernelGGL((mfma_k<16, true, 1>), dim3(blocks), dim3(256), 0, st, (float*)sink, iters, 1.0f);
  else if (shape == 162) hipLaunchKernelGGL((mfma_k<16, true, 2>), dim3(blocks), dim3(256), 0, st, (float*)sink, iters, 1.0f);
  else if (shape == 163) hipLaunchKernelGGL((mfma_k<16, true, 3>), dim3(blocks), dim3(256), 0, st, (float*)sink, iters, 1.0f);
  else if (shape == 16 && rnd) hipLaunchKernelGGL((mfma_k<16, true>), dim3(blocks), dim3(256), 0, st, (float*)sink, iters, 1.0f);
  else if (shape == 16) hipLaunchKernelGGL((mfma_k<16, false>), dim3(blocks), dim3(256), 0, st, (float*)sink, iters, 1.0f);
  else if (rnd) hipLaunchKernelGGL((mfma_k<32, true>), dim3(blocks), dim3(256), 0, st, (float*)sink, iters, 1.0f);
  else hipLaunchKernelGGL((mfma_k<32, false>), dim3(blocks), dim3(256), 0, st, (float*)sink, iters, 1.0f);
  return (int)hipGetLastError();
}
