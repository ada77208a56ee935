// Practical HBM read bandwidth of one MI355X: a grid-stride 16-byte-per-lane read of a buffer far larger than the 256 MB Infinity
// Cache, a few loads in flight per lane, result folded into one word per block so the loads cannot be dropped.
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/hbm_probe.hip -o tools/probes/hbm_probe.so ; python tools/probes/hbm_probe.py
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int UNR>
__global__ __launch_bounds__(256) void read_k(const uint4* __restrict__ p, int64_t n16, unsigned* __restrict__ sink) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  unsigned acc = 0;
  for (; i + (UNR - 1) * stride < n16; i += UNR * stride) {
    uint4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = p[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  for (; i < n16; i += stride) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) sink[blockIdx.x] = acc;  // (practically never: keeps the loads alive)
}

extern "C" int hbm_read(const void* p, int64_t bytes, void* sink, int blocks, int unr, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int64_t n16 = bytes / 16;
  if (unr == 8) hipLaunchKernelGGL(read_k<8>, dim3(blocks), dim3(256), 0, st, (const uint4*)p, n16, (unsigned*)sink);
  else if (unr == 4) hipLaunchKernelGGL(read_k<4>, dim3(blocks), dim3(256), 0, st, (const uint4*)p, n16, (unsigned*)sink);
  else hipLaunchKernelGGL(read_k<2>, dim3(blocks), dim3(256), 0, st, (const uint4*)p, n16, (unsigned*)sink);
  return (int)hipGetLastError();
}
