// see f8f6f4_probe.py
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__global__ void once_k(const int* a, const int* b, const int* sa, const int* sb, float* out) {
  const int l = threadIdx.x;
  i32x8 av, bv;
  for (int i = 0; i < 8; ++i) { av[i] = a[l * 8 + i]; bv[i] = b[l * 8 + i]; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, sa[l], 0, sb[l]);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}
extern "C" void run_once(const void* a, const void* b, const void* sa, const void* sb, void* out) {
  hipLaunchKernelGGL(once_k, dim3(1), dim3(64), 0, 0, (const int*)a, (const int*)b, (const int*)sa, (const int*)sb, (float*)out);
}

template <int WHICH>
__global__ __launch_bounds__(256) void rate_k(int n, float* sink) {
  i32x8 av, bv;
  for (int i = 0; i < 8; ++i) { av[i] = 0x38383838 + threadIdx.x; bv[i] = 0x38383838 ^ i; }
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const bf16x8 ha = __builtin_bit_cast(bf16x8, i32x4{av[0], av[1], av[2], av[3]}), hb = __builtin_bit_cast(bf16x8, i32x4{bv[0], bv[1], bv[2], bv[3]});
  f32x4 c[16];
  for (int i = 0; i < 16; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (WHICH) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c[i], 0, 0, 0, 127, 0, 127);
      else c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, c[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][3];
  if (s == 12345.678f) sink[0] = s;
}
static float* g_sink = nullptr;
extern "C" void run_rate(int which, int n) {
  if (!g_sink) hipMalloc(&g_sink, 16);
  if (which) hipLaunchKernelGGL(rate_k<1>, dim3(256), dim3(256), 0, 0, n, g_sink);
  else hipLaunchKernelGGL(rate_k<0>, dim3(256), dim3(256), 0, 0, n, g_sink);
}
