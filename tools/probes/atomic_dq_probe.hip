// What would the dQ accumulation of a FUSED attention backward cost on its own?  ("the 8.9 GB/layer of atomics was
// costed on paper, never measured".)  The fused form (dK, dV AND dQ from one pass over the scores: 5 MFMA products instead of 7) has
// every 128-key block add its contribution dS K to dQ[q, :] for all the queries it sees.  With the contraction over the block's 128 keys
// done inside the MFMA (each of the 4 waves takes a 32-wide d-slice of the 32 x 128 dQ tile of a 32-query half), a wave issues 16
// return-less global_atomic_add_f32 per half: lanes 0-31 / 32-63 cover 2 rows x 128 contiguous bytes.  This probe issues exactly that
// address stream for the cfg-3 geometry (B = 8, H = 32, S = 4096, D = 128, causal: 8.6 GB of atomics per layer call), all 128-key blocks
// of one (b, h) on one XCD as the attention kernels dispatch them (its 2 MB fp32 dQ slab then lives in that XCD's 4 MB L2), and nothing
// else - no MFMA work - so the time is the atomics' own.  Variants: atomics / plain stores of the same addresses / atomics with the
// queries visited in a per-block rotated order (fewer same-line collisions between blocks that run in step).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/atomic_dq_probe.hip -o /tmp/atomic_dq_probe && /tmp/atomic_dq_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

template <int MODE>  // 0 atomics, 1 plain stores, 2 atomics with rotated query order
__global__ __launch_bounds__(256) void dq_traffic_k(float* __restrict__ dq, int S, int H, int nchunk, int BH) {
  const int id = blockIdx.x, xcd = id & 7, idx = id >> 3;
  const int bl = idx / nchunk, chunk = idx % nchunk;
  const int bh = bl * 8 + xcd;
  if (bh >= BH) return;
  const int b = bh / H, h = bh % H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hi = lane >> 5;
  const int kv0 = chunk * 128;
  const int nhalf = (S - kv0) / 32;
  float* base = dq + ((int64_t)b * S) * (H * 128) + h * 128 + wave * 32 + l31;
  const int rot = MODE == 2 ? (chunk * 7) % nhalf : 0;
  for (int j = 0; j < nhalf; ++j) {
    int jj = j + rot;
    if (jj >= nhalf) jj -= nhalf;
    const int q0 = kv0 + jj * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      float* p = base + (int64_t)q * (H * 128);
      const float v = 1.0f + r;
      if (MODE == 1) __builtin_nontemporal_store(v, p);
      else __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

int main() {
  const int B = 8, H = 32, S = 4096, nchunk = S / 128, BH = B * H;
  const size_t n = (size_t)B * S * H * 128;
  float* dq;
  hipMalloc(&dq, n * 4);
  hipMemset(dq, 0, n * 4);
  const int grid = ((BH + 7) / 8) * 8 * nchunk;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  double bytes = 0;
  for (int c = 0; c < nchunk; ++c) bytes += (double)((S - c * 128) / 32) * 4 * 16 * 256;
  bytes *= BH;
  const char* names[3] = {"global_atomic_add_f32 (return-less)", "plain nontemporal stores, same addresses", "atomics, per-block rotated query order"};
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
      hipEventRecord(e0, 0);
      if (mode == 0) hipLaunchKernelGGL(dq_traffic_k<0>, dim3(grid), dim3(256), 0, 0, dq, S, H, nchunk, BH);
      if (mode == 1) hipLaunchKernelGGL(dq_traffic_k<1>, dim3(grid), dim3(256), 0, 0, dq, S, H, nchunk, BH);
      if (mode == 2) hipLaunchKernelGGL(dq_traffic_k<2>, dim3(grid), dim3(256), 0, 0, dq, S, H, nchunk, BH);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (it && ms < best) best = ms;
    }
    printf("%-48s %.3f ms  (%.2f GB -> %.2f TB/s)\n", names[mode], best, bytes / 1e9, bytes / best / 1e9);
  }
  // memset + convert pass the fused form also needs: 537 MB zero-fill, then 537 MB read + 268 MB write
  {
    hipEventRecord(e0, 0);
    hipMemsetAsync(dq, 0, n * 4, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-48s %.3f ms\n", "hipMemsetAsync of the fp32 dQ scratch (537 MB)", ms);
  }
  hipError_t e = hipGetLastError();
  printf("status: %s\n", hipGetErrorString(e));
  return 0;
}
