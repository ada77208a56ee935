// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds lds[i] = i (16-bit).  Each lane reads with
// byte address = lane * 8 (4 consecutive elements per lane) and we print what every lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds;
  unsigned addr;
  int l = threadIdx.x;
  if (mode == 0) addr = base + l * 8;                       // contiguous 8 B per lane
  else addr = base + ((l & 15) >> 2) * 256 + (l & 3) * 8 + (l >> 4) * 1024;  // 4 rows of 256 B per 16-lane group
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  out[l * 4 + 0] = v[0] & 0xffff; out[l * 4 + 1] = v[0] >> 16; out[l * 4 + 2] = v[1] & 0xffff; out[l * 4 + 3] = v[1] >> 16;
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("lane %2d: %4d %4d %4d %4d%s", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], (l % 4 == 3) ? "\n" : "   |   "); }
  }
  return 0;
}
