// HBM-bound elementwise kernels (gfx950): SwiGLU, quick-GELU, add, convert, RoPE, weight generator,
// CLIP im2col / embedding assembly, AdamW.  All use 16-byte (8 x 16-bit) accesses per lane and
// grid-stride loops capped at 256 CUs x 8 blocks.
#include "mh_common.h"

namespace {

inline int grid_for(int64_t nvec) {
  int64_t b = (nvec + 255) / 256;
  return (int)(b < 2048 ? (b > 0 ? b : 1) : 2048);
}


// out[t, f] = silu(gu[t, f]) * gu[t, ff + f]
template <int DT>
__global__ __launch_bounds__(256) void swiglu_fwd_k(const uint16_t* __restrict__ gu, uint16_t* __restrict__ out, int64_t rows, int ff) {
  const int64_t cpr = ff >> 3, total = rows * cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / cpr, c = i - t * cpr;
    const uint4* g = (const uint4*)(gu + t * 2 * ff) + c;
    const uint4* u = (const uint4*)(gu + t * 2 * ff + ff) + c;
    float a[8], b[8];
    unpack8<DT>(*g, a);
    unpack8<DT>(*u, b);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = swiglu_fwd1(a[k], b[k]);
    ((uint4*)(out + t * ff))[c] = pack8<DT>(a);
  }
}

template <int DT>
__global__ __launch_bounds__(256) void swiglu_bwd_k(const uint16_t* __restrict__ gu, const uint16_t* __restrict__ dout,
                                                    uint16_t* __restrict__ dgu, int64_t rows, int ff) {
  const int64_t cpr = ff >> 3, total = rows * cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / cpr, c = i - t * cpr;
    float a[8], b[8], d[8], dg[8], du[8];
    unpack8<DT>(((const uint4*)(gu + t * 2 * ff))[c], a);
    unpack8<DT>(((const uint4*)(gu + t * 2 * ff + ff))[c], b);
    unpack8<DT>(((const uint4*)(dout + t * ff))[c], d);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      swiglu_bwd1(a[k], b[k], d[k], dg[k], du[k]);
    }
    ((uint4*)(dgu + t * 2 * ff))[c] = pack8<DT>(dg);
    ((uint4*)(dgu + t * 2 * ff + ff))[c] = pack8<DT>(du);
  }
}

template <int DT, int MODE>  // 0: quick_gelu fwd (a=x), 1: quick_gelu bwd (a=x, b=dy), 2: add
__global__ __launch_bounds__(256) void ew2_k(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                                             uint16_t* __restrict__ y, int64_t nvec) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    float x[8], z[8];
    unpack8<DT>(((const uint4*)a)[i], x);
    if (MODE != 0) unpack8<DT>(((const uint4*)b)[i], z);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (MODE == 0) {
        x[k] = x[k] * sigmoidf_(1.702f * x[k]);
      } else if (MODE == 1) {
        const float s = sigmoidf_(1.702f * x[k]);
        x[k] = z[k] * s * (1.0f + 1.702f * x[k] * (1.0f - s));
      } else {
        x[k] = x[k] + z[k];
      }
    }
    ((uint4*)y)[i] = pack8<DT>(x);
  }
}

__device__ __forceinline__ float load_any(const void* p, int dt, int64_t i) {
  if (dt == MH_F32) return ((const float*)p)[i];
  if (dt == MH_BF16) return bf16_bits_to_f32(((const uint16_t*)p)[i]);
  return f16_bits_to_f32(((const uint16_t*)p)[i]);
}
__device__ __forceinline__ void store_any(void* p, int dt, int64_t i, float v) {
  if (dt == MH_F32) ((float*)p)[i] = v;
  else if (dt == MH_BF16) ((uint16_t*)p)[i] = (uint16_t)f32_to_bf16_bits(v);
  else ((uint16_t*)p)[i] = (uint16_t)f32_to_f16_bits(v);
}

__global__ __launch_bounds__(256) void convert_k(const void* __restrict__ src, int sdt, void* __restrict__ dst, int ddt, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    store_any(dst, ddt, i, load_any(src, sdt, i));
}

// ---- weight generator: must match merlin_amd/weights.py bit for bit -------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void fill_normal_k(void* __restrict__ out, int64_t n, uint64_t key, int64_t start,
                                                     float scale, float offset, int dt) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint64_t z = splitmix64(key + (uint64_t)(start + i));
    const int s = (int)((z & 0xffff) + ((z >> 16) & 0xffff) + ((z >> 32) & 0xffff) + (z >> 48)) - 131070;
    float v = __fmul_rn((float)s, scale);
    if (offset != 0.0f) v = __fadd_rn(v, offset);
    v = bf16_bits_to_f32(f32_to_bf16_bits(v));
    if (fabsf(v) < 6.103515625e-05f) v = 0.0f;
    store_any(out, dt, i, v);
  }
}

// ---- RoPE -------------------------------------------------------------------------------------
// table[s, j] = (cos(s * theta^(-2j/D)), sin(..)), j < D/2.  fp32, computed like the oracle: inv_freq in
// fp32, angle = s * inv_freq in fp32.
__global__ __launch_bounds__(256) void rope_table_k(float2* __restrict__ tab, int S, int D, float theta) {
  const int half = D >> 1;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= S * half) return;
  const int s = i / half, j = i - s * half;
  const float inv = 1.0f / powf(theta, (float)(2 * j) / (float)D);
  const float ang = (float)s * inv;
  tab[i] = make_float2(cosf(ang), sinf(ang));
}

// qkv [T, 3, H, D]; rotate q (which=0) and k (which=1) in place.  One thread handles 8 elements of the
// low half and the matching 8 of the high half.
template <int DT>
__global__ __launch_bounds__(256) void rope_qk_k(uint16_t* __restrict__ qkv, const float2* __restrict__ tab, int64_t T,
                                                 int S, int H, int D, int inverse) {
  const int half = D >> 1, vph = half >> 3;  // vectors per half-head
  const int64_t total = T * 2 * H * vph;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t r = i;
    const int v = (int)(r % vph); r /= vph;
    const int h = (int)(r % H); r /= H;
    const int which = (int)(r % 2);
    const int64_t t = r / 2;
    const int pos = (int)(t % S);
    uint16_t* base = qkv + ((t * 3 + which) * H + h) * (int64_t)D + v * 8;
    float lo[8], hi[8];
    unpack8<DT>(*(const uint4*)base, lo);
    unpack8<DT>(*(const uint4*)(base + half), hi);
    const float2* tb = tab + (int64_t)pos * half + v * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float c = tb[k].x, s = inverse ? -tb[k].y : tb[k].y;
      rope_rot(lo[k], hi[k], c, s, lo[k], hi[k]);  // x*cos + rotate_half(x)*sin, rotate_half = (-x2, x1)
    }
    *(uint4*)base = pack8<DT>(lo);
    *(uint4*)(base + half) = pack8<DT>(hi);
  }
}

// ---- CLIP patch embedding helpers -----------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void im2col_k(const void* __restrict__ pix, int pix_dt, uint16_t* __restrict__ cols,
                                                int N, int img, int ps, int Kpad, int rpi, int row0) {
  // image n owns rows [n*rpi, (n+1)*rpi): rows below row0 (the CLS slot) are zero, patch p sits at row n*rpi + row0 + p
  const int G = img / ps, K = 3 * ps * ps;
  const int64_t total = (int64_t)N * rpi * Kpad;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % Kpad);
    const int64_t row = i / Kpad;
    const int t = (int)(row % rpi) - row0;
    float v = 0.f;
    if (k < K && t >= 0 && t < G * G) {
      const int p = t;
      const int n = (int)(row / rpi);
      const int c = k / (ps * ps), rem = k - c * ps * ps, py = rem / ps, px = rem - py * ps;
      const int gy = p / G, gx = p - gy * G;
      v = load_any(pix, pix_dt, (((int64_t)n * 3 + c) * img + gy * ps + py) * img + gx * ps + px);
      // the reference casts pixels to the tower dtype before the conv (clip_encoder.py:76)
    }
    cols[i] = (uint16_t)st16<DT>(v);
  }
}

template <int DT>
__global__ __launch_bounds__(256) void vit_assemble_k(const uint16_t* __restrict__ patch, const uint16_t* __restrict__ cls,
                                                      const uint16_t* __restrict__ pos, uint16_t* __restrict__ x, int N,
                                                      int G2, int d) {
  // patch rows are laid out like x: [N, G2 + 1, d] with an (ignored) CLS slot per image
  const int vpr = d >> 3;
  const int64_t total = (int64_t)N * (G2 + 1) * vpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int v = (int)(i % vpr);
    const int64_t row = i / vpr;
    const int tkn = (int)(row % (G2 + 1));
    float a[8], b[8];
    if (tkn == 0) unpack8<DT>(((const uint4*)cls)[v], a);
    else unpack8<DT>(((const uint4*)(patch + row * d))[v], a);
    unpack8<DT>(((const uint4*)(pos + (int64_t)tkn * d))[v], b);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += b[k];
    ((uint4*)(x + row * d))[v] = pack8<DT>(a);
  }
}

// fp32 form (engine.fp32_residual: the tower's residual stream starts from fp32 values): patch rows are the fp32 output of the patch
// projection, x = fp32 [N, G2 + 1, d]
template <int DT>
__global__ __launch_bounds__(256) void vit_assemble_f32_k(const float* __restrict__ patch, const uint16_t* __restrict__ cls,
                                                          const uint16_t* __restrict__ pos, float* __restrict__ x, int N, int G2, int d) {
  const int vpr = d >> 3;
  const int64_t total = (int64_t)N * (G2 + 1) * vpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int v = (int)(i % vpr);
    const int64_t row = i / vpr;
    const int tkn = (int)(row % (G2 + 1));
    float a[8], b[8];
    if (tkn == 0) {
      unpack8<DT>(((const uint4*)cls)[v], a);
    } else {
      const float4 p0 = ((const float4*)(patch + row * d))[2 * v], p1 = ((const float4*)(patch + row * d))[2 * v + 1];
      a[0] = p0.x; a[1] = p0.y; a[2] = p0.z; a[3] = p0.w; a[4] = p1.x; a[5] = p1.y; a[6] = p1.z; a[7] = p1.w;
    }
    unpack8<DT>(((const uint4*)(pos + (int64_t)tkn * d))[v], b);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += b[k];
    ((float4*)(x + row * d))[2 * v] = make_float4(a[0], a[1], a[2], a[3]);
    ((float4*)(x + row * d))[2 * v + 1] = make_float4(a[4], a[5], a[6], a[7]);
  }
}

// dst[r, c] (=|+=) src[r, c] for a [rows, cols] block of two row-major 16-bit matrices with their own row strides (pad /
// un-pad of the patch-embedding weight and its gradient: widths that are not a multiple of 8 elements)
template <int DT>
__global__ __launch_bounds__(256) void copy2d_k(const uint16_t* __restrict__ src, int64_t lds_, uint16_t* __restrict__ dst, int64_t ldd,
                                                int rows, int cols, int accumulate) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    const uint16_t s = src[r * lds_ + c];
    uint16_t* d = dst + r * ldd + c;
    *d = accumulate ? (uint16_t)st16<DT>(ld16<DT>(*d) + ld16<DT>(s)) : s;
  }
}

// ---- AdamW (decoupled weight decay), 16-bit params/grads, fp32 moments --------------------------
template <int DT>
__global__ __launch_bounds__(256) void adamw_k(uint16_t* __restrict__ p, const uint16_t* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                               float wd, float bc1, float bc2, float gscale, const float* __restrict__ gscale_dev) {
  if (gscale_dev) gscale *= *gscale_dev;  // device-side factor (gradient clipping): no host round trip
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = ld16<DT>(g[i]) * gscale;
    float pi = ld16<DT>(p[i]);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    pi = pi * (1.f - lr * wd);
    pi -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
    p[i] = (uint16_t)st16<DT>(pi);
  }
}

template <int DT>
__global__ __launch_bounds__(256) void sumsq_k(const uint16_t* __restrict__ g, int64_t n, float* __restrict__ out) {
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float x = ld16<DT>(g[i]);
    s += x * x;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

// deterministic form: 16-byte loads, one partial per block (fixed grid), then a fixed-order final sum
template <int DT>
__global__ __launch_bounds__(256) void sumsq_part_k(const uint16_t* __restrict__ g, int64_t n, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  const int64_t nvec = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    float x[8];
    unpack8<DT>(((const uint4*)g)[i], x);
#pragma unroll
    for (int k = 0; k < 8; ++k) s = fmaf(x[k], x[k], s);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const float x = ld16<DT>(g[(nvec << 3) + threadIdx.x]);
    s = fmaf(x, x, s);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_k(const float* __restrict__ partial, int nblk, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace

#define DISPATCH16(dt, KERNEL, GRID, ...)                                                                       \
  do {                                                                                                          \
    if ((dt) == MH_BF16) hipLaunchKernelGGL(KERNEL<MH_BF16>, dim3(GRID), dim3(256), 0, as_stream(stream), __VA_ARGS__); \
    else if ((dt) == MH_F16) hipLaunchKernelGGL(KERNEL<MH_F16>, dim3(GRID), dim3(256), 0, as_stream(stream), __VA_ARGS__); \
    else return MH_ERR_DTYPE;                                                                                   \
  } while (0)

extern "C" int mh_swiglu_fwd(const void* gu, void* out, int rows, int ff, int dt, void* stream) {
  if (!gu || !out || rows <= 0 || (ff & 7)) return MH_ERR_ARG;
  DISPATCH16(dt, swiglu_fwd_k, grid_for((int64_t)rows * (ff >> 3)), (const uint16_t*)gu, (uint16_t*)out, (int64_t)rows, ff);
  MH_LAUNCH_CHECK();
}
extern "C" int mh_swiglu_bwd(const void* gu, const void* dout, void* dgu, int rows, int ff, int dt, void* stream) {
  if (!gu || !dout || !dgu || rows <= 0 || (ff & 7)) return MH_ERR_ARG;
  DISPATCH16(dt, swiglu_bwd_k, grid_for((int64_t)rows * (ff >> 3)), (const uint16_t*)gu, (const uint16_t*)dout, (uint16_t*)dgu, (int64_t)rows, ff);
  MH_LAUNCH_CHECK();
}
extern "C" int mh_quick_gelu_fwd(const void* x, void* y, int64_t n, int dt, void* stream) {
  if (!x || !y || n <= 0 || (n & 7)) return MH_ERR_ARG;
  if (dt == MH_BF16) hipLaunchKernelGGL((ew2_k<MH_BF16, 0>), dim3(grid_for(n >> 3)), dim3(256), 0, as_stream(stream), (const uint16_t*)x, (const uint16_t*)nullptr, (uint16_t*)y, n >> 3);
  else if (dt == MH_F16) hipLaunchKernelGGL((ew2_k<MH_F16, 0>), dim3(grid_for(n >> 3)), dim3(256), 0, as_stream(stream), (const uint16_t*)x, (const uint16_t*)nullptr, (uint16_t*)y, n >> 3);
  else return MH_ERR_DTYPE;
  MH_LAUNCH_CHECK();
}
extern "C" int mh_quick_gelu_bwd(const void* x, const void* dy, void* dx, int64_t n, int dt, void* stream) {
  if (!x || !dy || !dx || n <= 0 || (n & 7)) return MH_ERR_ARG;
  if (dt == MH_BF16) hipLaunchKernelGGL((ew2_k<MH_BF16, 1>), dim3(grid_for(n >> 3)), dim3(256), 0, as_stream(stream), (const uint16_t*)x, (const uint16_t*)dy, (uint16_t*)dx, n >> 3);
  else if (dt == MH_F16) hipLaunchKernelGGL((ew2_k<MH_F16, 1>), dim3(grid_for(n >> 3)), dim3(256), 0, as_stream(stream), (const uint16_t*)x, (const uint16_t*)dy, (uint16_t*)dx, n >> 3);
  else return MH_ERR_DTYPE;
  MH_LAUNCH_CHECK();
}
extern "C" int mh_add(const void* a, const void* b, void* y, int64_t n, int dt, void* stream) {
  if (!a || !b || !y || n <= 0 || (n & 7)) return MH_ERR_ARG;
  if (dt == MH_BF16) hipLaunchKernelGGL((ew2_k<MH_BF16, 2>), dim3(grid_for(n >> 3)), dim3(256), 0, as_stream(stream), (const uint16_t*)a, (const uint16_t*)b, (uint16_t*)y, n >> 3);
  else if (dt == MH_F16) hipLaunchKernelGGL((ew2_k<MH_F16, 2>), dim3(grid_for(n >> 3)), dim3(256), 0, as_stream(stream), (const uint16_t*)a, (const uint16_t*)b, (uint16_t*)y, n >> 3);
  else return MH_ERR_DTYPE;
  MH_LAUNCH_CHECK();
}
extern "C" int mh_convert(const void* src, int dt_src, void* dst, int dt_dst, int64_t n, void* stream) {
  if (!src || !dst || n <= 0) return MH_ERR_ARG;
  hipLaunchKernelGGL(convert_k, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), src, dt_src, dst, dt_dst, n);
  MH_LAUNCH_CHECK();
}
extern "C" int mh_fill_normal(void* out, int64_t n, uint64_t key, int64_t start, float sigma, float offset, int dt, void* stream) {
  if (!out || n <= 0) return MH_ERR_ARG;
  // identical to weights.scale_f32(): float32(sigma_double / std_double)
  const float scale = (float)((double)sigma / (65536.0 * sqrt(1.0 / 3.0)));
  hipLaunchKernelGGL(fill_normal_k, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), out, n, key, start, scale, offset, dt);
  MH_LAUNCH_CHECK();
}
extern "C" int mh_rope_table(float* cos_sin, int S, int D, float theta, void* stream) {
  if (!cos_sin || S <= 0 || (D & 15)) return MH_ERR_ARG;
  hipLaunchKernelGGL(rope_table_k, dim3((S * (D / 2) + 255) / 256), dim3(256), 0, as_stream(stream), (float2*)cos_sin, S, D, theta);
  MH_LAUNCH_CHECK();
}
extern "C" int mh_rope_qk(void* qkv, const float* cos_sin, int T, int S, int H, int D, int inverse, int dt, void* stream) {
  if (!qkv || !cos_sin || T <= 0 || (D & 15)) return MH_ERR_ARG;
  DISPATCH16(dt, rope_qk_k, grid_for((int64_t)T * 2 * H * (D / 16)), (uint16_t*)qkv, (const float2*)cos_sin, (int64_t)T, S, H, D, inverse);
  MH_LAUNCH_CHECK();
}
extern "C" int mh_im2col_patches(const void* pixels, int pix_dt, void* cols, int N, int img, int ps, int Kpad, int rows_per_img,
                                 int row0, int dt, void* stream) {
  if (!pixels || !cols || N <= 0 || img % ps != 0 || Kpad < 3 * ps * ps) return MH_ERR_ARG;
  const int G = img / ps;
  if (row0 < 0 || rows_per_img < row0 + G * G) return MH_ERR_ARG;
  DISPATCH16(dt, im2col_k, grid_for((int64_t)N * rows_per_img * Kpad), pixels, pix_dt, (uint16_t*)cols, N, img, ps, Kpad, rows_per_img, row0);
  MH_LAUNCH_CHECK();
}
extern "C" int mh_vit_assemble(const void* patch, const void* cls, const void* pos, void* x, int N, int G2, int d, int dt, void* stream) {
  if (!patch || !cls || !pos || !x || N <= 0 || (d & 7)) return MH_ERR_ARG;
  DISPATCH16(dt, vit_assemble_k, grid_for((int64_t)N * (G2 + 1) * (d >> 3)), (const uint16_t*)patch, (const uint16_t*)cls, (const uint16_t*)pos, (uint16_t*)x, N, G2, d);
  MH_LAUNCH_CHECK();
}
extern "C" int mh_vit_assemble_f32(const float* patch, const void* cls, const void* pos, float* x, int N, int G2, int d, int dt, void* stream) {
  if (!patch || !cls || !pos || !x || N <= 0 || (d & 7)) return MH_ERR_ARG;
  DISPATCH16(dt, vit_assemble_f32_k, grid_for((int64_t)N * (G2 + 1) * (d >> 3)), patch, (const uint16_t*)cls, (const uint16_t*)pos, x, N, G2, d);
  MH_LAUNCH_CHECK();
}
extern "C" int mh_copy2d(const void* src, int64_t lds, void* dst, int64_t ldd, int rows, int cols, int accumulate, int dt, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0 || lds < cols || ldd < cols) return MH_ERR_ARG;
  DISPATCH16(dt, copy2d_k, grid_for((int64_t)rows * cols), (const uint16_t*)src, lds, (uint16_t*)dst, ldd, rows, cols, accumulate);
  MH_LAUNCH_CHECK();
}
extern "C" int mh_adamw(void* p, const void* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                        float wd, int step, float gscale, int dt, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || step <= 0) return MH_ERR_ARG;
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  DISPATCH16(dt, adamw_k, grid_for(n), (uint16_t*)p, (const uint16_t*)g, m, v, n, lr, beta1, beta2, eps, wd, bc1, bc2, gscale, (const float*)nullptr);
  MH_LAUNCH_CHECK();
}
extern "C" int mh_adamw_clip(void* p, const void* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                             float wd, int step, float gscale, const float* gscale_dev, int dt, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || step <= 0) return MH_ERR_ARG;
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  DISPATCH16(dt, adamw_k, grid_for(n), (uint16_t*)p, (const uint16_t*)g, m, v, n, lr, beta1, beta2, eps, wd, bc1, bc2, gscale, gscale_dev);
  MH_LAUNCH_CHECK();
}
namespace {
__global__ void clip_scale_k(const float* __restrict__ sumsq, float gscale, float max_norm, float* __restrict__ out) {
  // torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1; total_norm of the SCALED gradients
  const float norm = sqrtf(sumsq[0]) * fabsf(gscale);
  out[0] = fminf(1.0f, max_norm / (norm + 1e-6f));
  out[1] = norm;
}
}  // namespace
extern "C" int mh_clip_scale(const float* sumsq, float gscale, float max_norm, float* out2, void* stream) {
  if (!sumsq || !out2 || max_norm <= 0.f) return MH_ERR_ARG;
  hipLaunchKernelGGL(clip_scale_k, dim3(1), dim3(1), 0, as_stream(stream), sumsq, gscale, max_norm, out2);
  MH_LAUNCH_CHECK();
}
namespace {
// flag |= 1 when any 16-bit element has a non-zero magnitude (bit pattern & 0x7fff: -0 is zero, NaN / Inf / denormals are NOT) - an EXACT test,
// unlike a sum of squares (fp32 squares of |g| < ~1e-23 flush to zero)
__global__ __launch_bounds__(256) void any_nonzero_k(const uint16_t* __restrict__ g, int64_t n, int* __restrict__ flag) {
  const int64_t n8 = n >> 3;
  unsigned acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 v = ((const uint4*)g)[i];
    acc |= (v.x | v.y | v.z | v.w) & 0x7fff7fffu;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 7)) acc |= g[(n8 << 3) + threadIdx.x] & 0x7fffu;
  if (__any(acc != 0) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
}  // namespace
extern "C" int mh_any_nonzero(const void* g, int64_t n, int* flag, void* stream) {
  if (!g || !flag || n <= 0 || ((uintptr_t)g & 15u)) return MH_ERR_ARG;
  hipLaunchKernelGGL(any_nonzero_k, dim3(grid_for(n >> 3)), dim3(256), 0, as_stream(stream), (const uint16_t*)g, n, flag);
  MH_LAUNCH_CHECK();
}
extern "C" int mh_sumsq(const void* g, int64_t n, float* out, int dt, void* stream) {
  if (!g || !out || n <= 0) return MH_ERR_ARG;
  DISPATCH16(dt, sumsq_k, grid_for(n), (const uint16_t*)g, n, out);
  MH_LAUNCH_CHECK();
}
extern "C" int mh_sumsq_det(const void* g, int64_t n, float* partial, float* out, int dt, void* stream) {
  if (!g || !partial || !out || n <= 0 || ((uintptr_t)g & 15u)) return MH_ERR_ARG;
  const int grid = grid_for(n >> 3);
  DISPATCH16(dt, sumsq_part_k, grid, (const uint16_t*)g, n, partial);
  hipLaunchKernelGGL(sumsq_final_k, dim3(1), dim3(256), 0, as_stream(stream), (const float*)partial, grid, out);
  MH_LAUNCH_CHECK();
}
