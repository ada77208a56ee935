// Does VALU / transcendental work of a wave run in the shadow of its own MFMAs?  (gfx950, one wave per SIMD)
// modes: 0 = 64 MFMA 32x32x16 per iteration; 1 = the VALU work alone; 2 = interleaved (one MFMA, then NV valu ops); per mode the VALU op is
// fma (full rate) or exp2 (transcendental).  Prints cycles per iteration from s_memtime and wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NV, bool TRANS>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, long long* cyc) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  u32x4 a = {0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, b = a;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = 0.001f * (threadIdx.x + i);
  const float s = 0.9999f, t = 1e-6f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 64; ++m) {
      if (MODE != 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[m & 3]) : "v"(a), "v"(b));
      if (MODE != 0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          if (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(m * NV + v) & 7]));
          else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(m * NV + v) & 7]) : "v"(s), "v"(t));
        }
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]));
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += acc[i][0];
  for (int i = 0; i < 8; ++i) r += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE, int NV, bool TRANS>
void run(const char* name, float* out, long long* cyc) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, NV, TRANS><<<256, 256>>>(out, 10, cyc);
  hipEventRecord(e0);
  k<MODE, NV, TRANS><<<256, 256>>>(out, iters, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s %8.3f ms  %8.1f ns/iter  %8.1f counter ticks/iter\n", name, ms, ms * 1e6 / iters, (double)c / iters);
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  run<0, 0, false>("64 MFMA", out, cyc);
  run<1, 4, false>("64 x 4 fma", out, cyc);
  run<2, 4, false>("64 x (MFMA + 4 fma)", out, cyc);
  run<1, 7, false>("64 x 7 fma", out, cyc);
  run<2, 7, false>("64 x (MFMA + 7 fma)", out, cyc);
  run<1, 1, true>("64 x 1 exp", out, cyc);
  run<2, 1, true>("64 x (MFMA + 1 exp)", out, cyc);
  run<1, 2, true>("64 x 2 exp", out, cyc);
  run<2, 2, true>("64 x (MFMA + 2 exp)", out, cyc);
  return 0;
}
