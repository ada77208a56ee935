// ConvProjector support (mmgpt/model/projector/conv_projector.py:23-39): Conv2d(C -> d, k=3, stride s, pad 1)
// over the G x G patch grid, run as an implicit GEMM: a gather builds cols[(n,oy,ox), (c,ky,kx)] (the k order of
// `weight.view(d, C*9)`, so the weight is used in place) for the MFMA GEMM; the backward scatter is written as a
// GATHER (each input element sums the <= 9 output taps that touched it), so it is deterministic and atomic-free.
// x is the tower output WITH its CLS rows ([N*(G2+1), C], patch p of image n at row n*(G2+1) + 1 + p).
#include "mh_common.h"

namespace {

inline int grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 4096 ? (b > 0 ? b : 1) : 4096);
}

__global__ __launch_bounds__(256) void conv3x3_cols_k(const uint16_t* __restrict__ x, uint16_t* __restrict__ cols, int N, int G,
                                                      int C, int stride, int rows_per_img, int row0) {
  const int Go = (G + 2 - 3) / stride + 1;
  const int64_t K = (int64_t)C * 9;
  const int64_t total = (int64_t)N * Go * Go * K;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t k = i % K, row = i / K;
    const int c = (int)(k / 9), tap = (int)(k % 9), ky = tap / 3, kx = tap % 3;
    const int ox = (int)(row % Go), oy = (int)((row / Go) % Go), n = (int)(row / ((int64_t)Go * Go));
    const int iy = oy * stride + ky - 1, ix = ox * stride + kx - 1;
    uint16_t v = 0;
    if (iy >= 0 && iy < G && ix >= 0 && ix < G) v = x[((int64_t)n * rows_per_img + row0 + iy * G + ix) * C + c];
    cols[i] = v;
  }
}

// dx[n, row0 + iy*G + ix, c] = sum over taps of dcols[(n, oy, ox), c*9 + ky*3 + kx] with iy = oy*s + ky - 1 ...
// rows outside the patch grid (CLS) are written as zero.
template <int DT>
__global__ __launch_bounds__(256) void conv3x3_col2im_k(const uint16_t* __restrict__ dcols, uint16_t* __restrict__ dx, int N, int G,
                                                        int C, int stride, int rows_per_img, int row0) {
  const int Go = (G + 2 - 3) / stride + 1;
  const int64_t K = (int64_t)C * 9;
  const int64_t total = (int64_t)N * rows_per_img * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t r = i / C;
    const int pr = (int)(r % rows_per_img) - row0, n = (int)(r / rows_per_img);
    float acc = 0.f;
    if (pr >= 0 && pr < G * G) {
      const int iy = pr / G, ix = pr % G;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int ty = iy + 1 - ky;
        if (ty < 0 || ty % stride != 0) continue;
        const int oy = ty / stride;
        if (oy >= Go) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int tx = ix + 1 - kx;
          if (tx < 0 || tx % stride != 0) continue;
          const int ox = tx / stride;
          if (ox >= Go) continue;
          acc += ld16<DT>(dcols[(((int64_t)n * Go + oy) * Go + ox) * K + c * 9 + ky * 3 + kx]);
        }
      }
    }
    dx[i] = (uint16_t)st16<DT>(acc);
  }
}

}  // namespace

extern "C" int mh_conv3x3_cols(const void* x, void* cols, int N, int G, int C, int stride, int rows_per_img, int row0, void* stream) {
  if (!x || !cols || N <= 0 || G <= 0 || C <= 0 || stride <= 0 || rows_per_img < row0 + G * G) return MH_ERR_ARG;
  const int Go = (G + 2 - 3) / stride + 1;
  hipLaunchKernelGGL(conv3x3_cols_k, dim3(grid_for((int64_t)N * Go * Go * C * 9)), dim3(256), 0, as_stream(stream),
                     (const uint16_t*)x, (uint16_t*)cols, N, G, C, stride, rows_per_img, row0);
  MH_LAUNCH_CHECK();
}

extern "C" int mh_conv3x3_col2im(const void* dcols, void* dx, int N, int G, int C, int stride, int rows_per_img, int row0, int dt,
                                 void* stream) {
  if (!dcols || !dx || N <= 0 || G <= 0 || C <= 0 || stride <= 0 || rows_per_img < row0 + G * G) return MH_ERR_ARG;
  const int grid = grid_for((int64_t)N * rows_per_img * C);
  if (dt == MH_BF16)
    hipLaunchKernelGGL(conv3x3_col2im_k<MH_BF16>, dim3(grid), dim3(256), 0, as_stream(stream), (const uint16_t*)dcols, (uint16_t*)dx, N, G, C, stride, rows_per_img, row0);
  else if (dt == MH_F16)
    hipLaunchKernelGGL(conv3x3_col2im_k<MH_F16>, dim3(grid), dim3(256), 0, as_stream(stream), (const uint16_t*)dcols, (uint16_t*)dx, N, G, C, stride, rows_per_img, row0);
  else return MH_ERR_DTYPE;
  MH_LAUNCH_CHECK();
}
