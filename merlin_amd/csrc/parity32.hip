// fp32-STORE parity mode (SURVEY §8d cfg 2: "an fp16/fp32-accumulate mode meeting 1e-3"): the same forward
// with every activation kept in fp32 in HBM, so that the comparison with the reference's fp32 CPU path measures the KERNELS'
// arithmetic and not 16-bit storage (which alone costs 1.2e-3 .. 4.6e-3 at real widths / full depth,
// profiles/r01_full_depth_rounding_attribution.txt).  Not a performance path.
//
// Linear layers stay on the bf16 MFMA GEMM: an fp32 activation is split EXACTLY into three bf16 terms x = hi + mid + lo
// (8 + 8 + 8 mantissa bits; bf16 has fp32's exponent range, so no term underflows), the weights are exactly representable in
// bf16 (merlin_amd/weights.py), hence x W^T = hi W^T + mid W^T + lo W^T with exact products and fp32 accumulation: three launches
// of the production kernel with fp32 output + accumulate.  Everything else is restated here as plain fp32 VALU kernels:
// norms, RoPE, SwiGLU / quick-GELU, residual adds, embedding / splice, patch im2col + assembly, conv gather, and a
// straightforward (one wave per query row, online softmax) attention.
#include "mh_common.h"

namespace {

inline int grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 4096 ? (b > 0 ? b : 1) : 4096);
}

__global__ __launch_bounds__(256) void split3_k(const float* __restrict__ x, uint16_t* __restrict__ hi, uint16_t* __restrict__ mid,
                                                uint16_t* __restrict__ lo, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    const uint32_t h = f32_to_bf16_bits(v);
    const float r1 = v - bf16_bits_to_f32(h);  // exact
    const uint32_t m = f32_to_bf16_bits(r1);
    const float r2 = r1 - bf16_bits_to_f32(m);  // exact
    hi[i] = (uint16_t)h;
    mid[i] = (uint16_t)m;
    lo[i] = (uint16_t)f32_to_bf16_bits(r2);
  }
}

// one wave per row; weights bf16
__global__ __launch_bounds__(256) void rmsnorm32_k(const float* __restrict__ x, const uint16_t* __restrict__ w, float* __restrict__ y, int rows,
                                                   int d, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (int64_t)row * d;
  float ss = 0.f;
  for (int i = lane; i < d; i += 64) ss += xr[i] * xr[i];
  ss = wave_sum(ss);
  const float r = rsqrtf(ss / (float)d + eps);
  for (int i = lane; i < d; i += 64) y[(int64_t)row * d + i] = bf16_bits_to_f32(w[i]) * (xr[i] * r);
}

__global__ __launch_bounds__(256) void layernorm32_k(const float* __restrict__ x, const uint16_t* __restrict__ w, const uint16_t* __restrict__ b,
                                                     float* __restrict__ y, int rows, int d, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (int64_t)row * d;
  float s = 0.f;
  for (int i = lane; i < d; i += 64) s += xr[i];
  const float mean = wave_sum(s) / (float)d;
  float v = 0.f;
  for (int i = lane; i < d; i += 64) { const float c = xr[i] - mean; v += c * c; }
  const float r = rsqrtf(wave_sum(v) / (float)d + eps);
  for (int i = lane; i < d; i += 64) y[(int64_t)row * d + i] = (xr[i] - mean) * r * bf16_bits_to_f32(w[i]) + bf16_bits_to_f32(b[i]);
}

// op 0: y = a + b;  1: y[t, f] = silu(a[t, f]) * a[t, ff + f] (a = gate|up rows of width 2*ff, n = rows*ff);  2: y = quick_gelu(a)
__global__ __launch_bounds__(256) void ew32_k(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n, int op,
                                              int ff) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    if (op == 0) y[i] = a[i] + b[i];
    else if (op == 1) {
      const int64_t t = i / ff, f = i - t * ff;
      const float g = a[t * 2 * ff + f], u = a[t * 2 * ff + ff + f];
      y[i] = g / (1.0f + expf(-g)) * u;
    } else {
      const float v = a[i];
      y[i] = v / (1.0f + expf(-1.702f * v));
    }
  }
}

// qkv [T, 3, H, D] fp32: rotate q and k (rotate-half), position t % S
__global__ __launch_bounds__(256) void rope32_k(float* __restrict__ qkv, const float2* __restrict__ tab, int64_t T, int S, int H, int D) {
  const int half = D >> 1;
  const int64_t total = T * 2 * H * half;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t r = i;
    const int j = (int)(r % half); r /= half;
    const int h = (int)(r % H); r /= H;
    const int which = (int)(r % 2);
    const int64_t t = r / 2;
    float* base = qkv + ((t * 3 + which) * H + h) * (int64_t)D;
    const float2 cs = tab[(int64_t)(t % S) * half + j];
    const float a = base[j], b = base[j + half];
    base[j] = a * cs.x - b * cs.y;
    base[j + half] = b * cs.x + a * cs.y;
  }
}

// out[t] = src[t] >= 0 ? feats[src[t]] : embed[ids[t]]  (embed bf16, feats fp32)
__global__ __launch_bounds__(256) void embed_splice32_k(const int64_t* __restrict__ ids, const int32_t* __restrict__ src, const uint16_t* __restrict__ embed,
                                                        const float* __restrict__ feats, float* __restrict__ out, int64_t T, int d) {
  const int64_t total = T * d;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / d;
    const int c = (int)(i - t * d);
    const int s = src ? src[t] : -1;
    out[i] = s >= 0 ? feats[(int64_t)s * d + c] : bf16_bits_to_f32(embed[ids[t] * (int64_t)d + c]);
  }
}

__global__ __launch_bounds__(256) void im2col32_k(const float* __restrict__ pix, float* __restrict__ cols, int N, int img, int ps, int Kpad, int rpi,
                                                  int row0) {
  const int G = img / ps, K = 3 * ps * ps;
  const int64_t total = (int64_t)N * rpi * Kpad;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % Kpad);
    const int64_t row = i / Kpad;
    const int t = (int)(row % rpi) - row0;
    float v = 0.f;
    if (k < K && t >= 0 && t < G * G) {
      const int n = (int)(row / rpi);
      const int c = k / (ps * ps), rem = k - c * ps * ps, py = rem / ps, px = rem - py * ps;
      const int gy = t / G, gx = t - gy * G;
      v = pix[(((int64_t)n * 3 + c) * img + gy * ps + py) * img + gx * ps + px];
    }
    cols[i] = v;
  }
}

__global__ __launch_bounds__(256) void vit_assemble32_k(const float* __restrict__ patch, const uint16_t* __restrict__ cls, const uint16_t* __restrict__ pos,
                                                        float* __restrict__ x, int N, int G2, int d) {
  const int64_t total = (int64_t)N * (G2 + 1) * d;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % d);
    const int64_t row = i / d;
    const int tkn = (int)(row % (G2 + 1));
    const float a = tkn == 0 ? bf16_bits_to_f32(cls[c]) : patch[i];
    x[i] = a + bf16_bits_to_f32(pos[(int64_t)tkn * d + c]);
  }
}

__global__ __launch_bounds__(256) void conv3x3_cols32_k(const float* __restrict__ x, float* __restrict__ cols, int N, int G, int C, int stride,
                                                        int rows_per_img, int row0) {
  const int Go = (G + 2 - 3) / stride + 1;
  const int64_t K = (int64_t)C * 9;
  const int64_t total = (int64_t)N * Go * Go * K;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t k = i % K, row = i / K;
    const int c = (int)(k / 9), tap = (int)(k % 9), ky = tap / 3, kx = tap % 3;
    const int ox = (int)(row % Go), oy = (int)((row / Go) % Go), n = (int)(row / ((int64_t)Go * Go));
    const int iy = oy * stride + ky - 1, ix = ox * stride + kx - 1;
    cols[i] = (iy >= 0 && iy < G && ix >= 0 && ix < G) ? x[((int64_t)n * rows_per_img + row0 + iy * G + ix) * C + c] : 0.f;
  }
}

// One wave per (b, h, query row); lanes split D.  Keys [0, min(len, q + 1) if causal else len); rows >= len give zeros.
template <int D>
__global__ __launch_bounds__(256) void attn32_k(const float* __restrict__ q, int64_t ldq, const float* __restrict__ k, int64_t ldk,
                                                const float* __restrict__ v, int64_t ldv, float* __restrict__ o, int64_t ldo,
                                                const int32_t* __restrict__ lens, int B, int S, int H, int causal) {
  constexpr int E = D / 64;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (w >= (int64_t)B * H * S) return;
  const int qi = (int)(w % S);
  const int h = (int)((w / S) % H);
  const int b = (int)(w / ((int64_t)S * H));
  const int len = lens ? min(lens[b], S) : S;
  float* op = o + ((int64_t)b * S + qi) * ldo + (int64_t)h * D;
  if (qi >= len) {
#pragma unroll
    for (int e = 0; e < E; ++e) op[lane + 64 * e] = 0.f;
    return;
  }
  const float scale = rsqrtf((float)D);
  float qv[E], acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    qv[e] = q[((int64_t)b * S + qi) * ldq + (int64_t)h * D + lane + 64 * e] * scale;
    acc[e] = 0.f;
  }
  const int kv_end = causal ? min(len, qi + 1) : len;
  float m = -INFINITY, l = 0.f;
  for (int j = 0; j < kv_end; ++j) {
    const float* kp = k + ((int64_t)b * S + j) * ldk + (int64_t)h * D;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) s += qv[e] * kp[lane + 64 * e];
    s = wave_sum(s);
    const float mn = fmaxf(m, s);
    const float corr = expf(m - mn), p = expf(s - mn);
    l = l * corr + p;
    const float* vp = v + ((int64_t)b * S + j) * ldv + (int64_t)h * D;
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = acc[e] * corr + p * vp[lane + 64 * e];
    m = mn;
  }
  const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) op[lane + 64 * e] = acc[e] * inv;
}

}  // namespace

#define P32_LAUNCH(K, GRID, ...) hipLaunchKernelGGL(K, dim3(GRID), dim3(256), 0, as_stream(stream), __VA_ARGS__)

extern "C" int mh_p32_split3(const float* x, void* hi, void* mid, void* lo, int64_t n, void* stream) {
  if (!x || !hi || !mid || !lo || n <= 0) return MH_ERR_ARG;
  P32_LAUNCH(split3_k, grid_for(n), x, (uint16_t*)hi, (uint16_t*)mid, (uint16_t*)lo, n);
  MH_LAUNCH_CHECK();
}
extern "C" int mh_p32_rmsnorm(const float* x, const void* w_bf16, float* y, int rows, int d, float eps, void* stream) {
  if (!x || !w_bf16 || !y || rows <= 0 || d <= 0) return MH_ERR_ARG;
  P32_LAUNCH(rmsnorm32_k, (rows + 3) / 4, x, (const uint16_t*)w_bf16, y, rows, d, eps);
  MH_LAUNCH_CHECK();
}
extern "C" int mh_p32_layernorm(const float* x, const void* w_bf16, const void* b_bf16, float* y, int rows, int d, float eps, void* stream) {
  if (!x || !w_bf16 || !b_bf16 || !y || rows <= 0 || d <= 0) return MH_ERR_ARG;
  P32_LAUNCH(layernorm32_k, (rows + 3) / 4, x, (const uint16_t*)w_bf16, (const uint16_t*)b_bf16, y, rows, d, eps);
  MH_LAUNCH_CHECK();
}
extern "C" int mh_p32_elementwise(const float* a, const float* b, float* y, int64_t n, int op, int ff, void* stream) {
  if (!a || !y || n <= 0 || op < 0 || op > 2 || (op == 0 && !b) || (op == 1 && ff <= 0)) return MH_ERR_ARG;
  P32_LAUNCH(ew32_k, grid_for(n), a, b, y, n, op, ff);
  MH_LAUNCH_CHECK();
}
extern "C" int mh_p32_rope(float* qkv, const float* cos_sin, int64_t T, int S, int H, int D, void* stream) {
  if (!qkv || !cos_sin || T <= 0 || (D & 1)) return MH_ERR_ARG;
  P32_LAUNCH(rope32_k, grid_for(T * 2 * H * (D / 2)), qkv, (const float2*)cos_sin, T, S, H, D);
  MH_LAUNCH_CHECK();
}
extern "C" int mh_p32_embed_splice(const int64_t* ids, const int32_t* src, const void* embed_bf16, const float* feats, float* out, int64_t T, int d,
                                   void* stream) {
  if (!ids || !embed_bf16 || !out || T <= 0 || d <= 0 || (src && !feats)) return MH_ERR_ARG;
  P32_LAUNCH(embed_splice32_k, grid_for(T * d), ids, src, (const uint16_t*)embed_bf16, feats, out, T, d);
  MH_LAUNCH_CHECK();
}
extern "C" int mh_p32_im2col(const float* pixels, float* cols, int N, int img, int ps, int Kpad, int rows_per_img, int row0, void* stream) {
  if (!pixels || !cols || N <= 0 || img % ps != 0 || Kpad < 3 * ps * ps) return MH_ERR_ARG;
  P32_LAUNCH(im2col32_k, grid_for((int64_t)N * rows_per_img * Kpad), pixels, cols, N, img, ps, Kpad, rows_per_img, row0);
  MH_LAUNCH_CHECK();
}
extern "C" int mh_p32_vit_assemble(const float* patch, const void* cls_bf16, const void* pos_bf16, float* x, int N, int G2, int d, void* stream) {
  if (!patch || !cls_bf16 || !pos_bf16 || !x || N <= 0) return MH_ERR_ARG;
  P32_LAUNCH(vit_assemble32_k, grid_for((int64_t)N * (G2 + 1) * d), patch, (const uint16_t*)cls_bf16, (const uint16_t*)pos_bf16, x, N, G2, d);
  MH_LAUNCH_CHECK();
}
extern "C" int mh_p32_conv3x3_cols(const float* x, float* cols, int N, int G, int C, int stride, int rows_per_img, int row0, void* stream) {
  if (!x || !cols || N <= 0 || G <= 0 || C <= 0 || stride <= 0 || rows_per_img < row0 + G * G) return MH_ERR_ARG;
  const int Go = (G + 2 - 3) / stride + 1;
  P32_LAUNCH(conv3x3_cols32_k, grid_for((int64_t)N * Go * Go * C * 9), x, cols, N, G, C, stride, rows_per_img, row0);
  MH_LAUNCH_CHECK();
}
extern "C" int mh_p32_attention(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, float* o, int64_t ldo,
                                const int32_t* seqlens, int B, int S, int H, int D, int causal, void* stream) {
  if (!q || !k || !v || !o || B <= 0 || S <= 0 || H <= 0) return MH_ERR_ARG;
  const int64_t nw = (int64_t)B * H * S;
  const unsigned grid = (unsigned)((nw + 3) / 4);
  if (D == 128) P32_LAUNCH(attn32_k<128>, grid, q, ldq, k, ldk, v, ldv, o, ldo, seqlens, B, S, H, causal);
  else if (D == 64) P32_LAUNCH(attn32_k<64>, grid, q, ldq, k, ldk, v, ldv, o, ldo, seqlens, B, S, H, causal);
  else return MH_ERR_SHAPE;
  MH_LAUNCH_CHECK();
}
