// RMSNorm / LayerNorm forward + backward (gfx950).  HBM-bound row kernels:
//   one wave (64 lanes) per row, 4 rows per 256-thread block, 16-byte (8 x 16-bit) loads per lane,
//   fp32 statistics reduced with wave shuffles only (no LDS, no barrier in the forward).
// Backward: dx per row + per-block fp32 partial sums of dw (and db) kept in registers across the
// block's rows, written once as partial[nblk, d] and finished by mh_reduce_partials (deterministic,
// no atomics).
// Reference arithmetic: transformers LlamaRMSNorm (modeling_llama.py: w * x * rsqrt(mean(x^2)+eps))
// and nn.LayerNorm (CLIP pre_layrnorm / layer_norm1/2, eps 1e-5).
#include "mh_common.h"

namespace {

#define TOUCH4(q_) asm volatile("" : "+v"((q_).x), "+v"((q_).y), "+v"((q_).z), "+v"((q_).w))
constexpr int ROWS_PER_BLOCK = 4;
constexpr int MAX_PARTIALS = 1024;

template <int DT>
__global__ __launch_bounds__(256) void rmsnorm_fwd_k(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                     uint16_t* __restrict__ y, float* __restrict__ rstd_out,
                                                     int rows, int d, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * ROWS_PER_BLOCK + wave;
  if (row >= rows) return;
  const uint4* xr = (const uint4*)(x + (int64_t)row * d);
  const uint4* wr = (const uint4*)w;
  uint4* yr = (uint4*)(y + (int64_t)row * d);
  const int nch = d >> 3;
  float ss = 0.f;
  for (int c = lane; c < nch; c += 64) {
    float f[8];
    unpack8<DT>(xr[c], f);
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += f[i] * f[i];
  }
  ss = wave_sum(ss);
  const float r = rsqrtf(ss / (float)d + eps);
  if (rstd_out && lane == 0) rstd_out[row] = r;
  for (int c = lane; c < nch; c += 64) {
    float f[8], g[8];
    unpack8<DT>(xr[c], f);
    unpack8<DT>(wr[c], g);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = f[i] * r * g[i];
    yr[c] = pack8<DT>(f);
  }
}

// A few rows only (the decode step: 1-16 tokens): one wave per row leaves the row's 8 KB to 64 lanes in eight dependent round
// trips.  Here a BLOCK takes a row - every thread requests its (up to four) 16-byte chunks at once - and the block-wide sum is
// formed in a fixed order (wave sums, then waves 0..3), so the result is deterministic; it may differ from rmsnorm_fwd_k's by the
// rounding of the sum.
template <int DT>
__global__ __launch_bounds__(256) void rmsnorm_fwd_row_k(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                         uint16_t* __restrict__ y, float* __restrict__ rstd_out, int d, float eps) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const uint4* xr = (const uint4*)(x + (int64_t)row * d);
  const uint4* wr = (const uint4*)w;
  uint4* yr = (uint4*)(y + (int64_t)row * d);
  const int nch = d >> 3;  // <= 1024 (checked by the launcher)
  uint4 xv[4], wv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + i * 256;
    xv[i] = c < nch ? xr[c] : make_uint4(0, 0, 0, 0);
    wv[i] = c < nch ? wr[c] : make_uint4(0, 0, 0, 0);
  }
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float f[8];
    unpack8<DT>(xv[i], f);
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += f[e] * f[e];
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  ss = (red[0] + red[1]) + (red[2] + red[3]);
  const float r = rsqrtf(ss / (float)d + eps);
  if (rstd_out && tid == 0) rstd_out[row] = r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + i * 256;
    if (c < nch) {
      float f[8], g[8];
      unpack8<DT>(xv[i], f);
      unpack8<DT>(wv[i], g);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = f[e] * r * g[e];
      yr[c] = pack8<DT>(f);
    }
  }
}

// RMSNorm forward that also emits the row-quantised e4m3 copy of its (16-bit rounded) output + the row scale: the operand of the
// fp8 GEMM that follows, without a separate pass over y (BASELINE cfg 5: quantisation fused into the norm).  Bit-identical to
// rmsnorm_fwd_k followed by quant_fp8_rows_k.  x and w are re-read from L1/L2 (a row is 8 KB), y and q are written once.
template <int DT, int NCH = 0>
__global__ __launch_bounds__(256) void rmsnorm_fwd_q8_k(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, uint16_t* __restrict__ y,
                                                        uint8_t* __restrict__ q, float* __restrict__ sc, int rows, int d, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * ROWS_PER_BLOCK + wave;
  if (row >= rows) return;
  const uint4* xr = (const uint4*)(x + (int64_t)row * d);
  const uint4* wr = (const uint4*)w;
  uint4* yr = (uint4*)(y + (int64_t)row * d);
  const int nch = d >> 3;
  if constexpr (NCH > 0) {
    // d == 512 * NCH: x and w requested once, up front; the rounded outputs stay packed in registers between the passes
    uint4 xq[NCH], wq[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      xq[j] = xr[lane + 64 * j];
      wq[j] = wr[lane + 64 * j];
    }
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      float f[8];
      unpack8<DT>(xq[j], f);
#pragma unroll
      for (int i = 0; i < 8; ++i) ss += f[i] * f[i];
    }
    ss = wave_sum(ss);
    const float r = rsqrtf(ss / (float)d + eps);
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      float f[8], g[8];
      unpack8<DT>(xq[j], f);
      unpack8<DT>(wq[j], g);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = f[i] * r * g[i];
      const uint4 pk = pack8<DT>(f);
      yr[lane + 64 * j] = pk;
      xq[j] = pk;  // the ROUNDED values are what the stand-alone quantiser sees
      unpack8<DT>(pk, f);
#pragma unroll
      for (int i = 0; i < 8; ++i) mx = fmaxf(mx, fabsf(f[i]));
    }
    mx = wave_max(mx);
    const float s = mx > 0.f ? mx * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / s;
    if (lane == 0) sc[row] = s;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      float f[8];
      unpack8<DT>(xq[j], f);
      int p0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, 0, false);
      p0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, p0, true);
      int p1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, 0, false);
      p1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, p1, true);
      *(uint2*)(q + (int64_t)row * d + (lane + 64 * j) * 8) = make_uint2((unsigned)p0, (unsigned)p1);
    }
    return;
  }
  float ss = 0.f;
  for (int c = lane; c < nch; c += 64) {
    float f[8];
    unpack8<DT>(xr[c], f);
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += f[i] * f[i];
  }
  ss = wave_sum(ss);
  const float r = rsqrtf(ss / (float)d + eps);
  float mx = 0.f;
  for (int c = lane; c < nch; c += 64) {
    float f[8], g[8];
    unpack8<DT>(xr[c], f);
    unpack8<DT>(wr[c], g);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = f[i] * r * g[i];
    const uint4 pk = pack8<DT>(f);
    yr[c] = pk;
    unpack8<DT>(pk, f);  // the ROUNDED values are what the stand-alone quantiser sees
#pragma unroll
    for (int i = 0; i < 8; ++i) mx = fmaxf(mx, fabsf(f[i]));
  }
  mx = wave_max(mx);
  const float s = mx > 0.f ? mx * (1.0f / 448.0f) : 1.0f;
  const float inv = 1.0f / s;
  if (lane == 0) sc[row] = s;
  for (int c = lane; c < nch; c += 64) {
    float f[8], g[8];
    unpack8<DT>(xr[c], f);
    unpack8<DT>(wr[c], g);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = f[i] * r * g[i];
    unpack8<DT>(pack8<DT>(f), f);
    int p0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, 0, false);
    p0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, p0, true);
    int p1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, 0, false);
    p1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, p1, true);
    *(uint2*)(q + (int64_t)row * d + c * 8) = make_uint2((unsigned)p0, (unsigned)p1);
  }
}

template <int DT>
__global__ __launch_bounds__(256) void layernorm_fwd_k(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                       const uint16_t* __restrict__ b, uint16_t* __restrict__ y,
                                                       int rows, int d, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * ROWS_PER_BLOCK + wave;
  if (row >= rows) return;
  const uint4* xr = (const uint4*)(x + (int64_t)row * d);
  const uint4* wr = (const uint4*)w;
  const uint4* br = (const uint4*)b;
  uint4* yr = (uint4*)(y + (int64_t)row * d);
  const int nch = d >> 3;
  float s = 0.f;
  for (int c = lane; c < nch; c += 64) {
    float f[8];
    unpack8<DT>(xr[c], f);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += f[i];
  }
  const float mu = wave_sum(s) / (float)d;
  float ss = 0.f;
  for (int c = lane; c < nch; c += 64) {
    float f[8];
    unpack8<DT>(xr[c], f);
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += (f[i] - mu) * (f[i] - mu);
  }
  const float r = rsqrtf(wave_sum(ss) / (float)d + eps);
  for (int c = lane; c < nch; c += 64) {
    float f[8], g[8], h[8];
    unpack8<DT>(xr[c], f);
    unpack8<DT>(wr[c], g);
    unpack8<DT>(br[c], h);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (f[i] - mu) * r * g[i] + h[i];
    yr[c] = pack8<DT>(f);
  }
}

// fp32-RESIDUAL-STREAM forms (engine.fp32_residual; HISTORY.md §4 "fp32 residual streams"): the stream x is fp32 [rows, d] and is
// updated in place by the o / down (out_proj / fc2) projections' accumulating fp32 epilogue, so the only 16-bit roundings left on
// the forward path are the GEMM operands.  This kernel is the stream's reader: y = norm(x) in 16 bits for the next GEMM, and
// (x16 != NULL) the 16-bit copy of x that the backward keeps as the layer input.  Same arithmetic and summation order as
// rmsnorm_fwd_k / layernorm_fwd_k; LN = true: LayerNorm (b != NULL).
// Y32 = true: y is fp32 (the CLIP tower's pre_layrnorm writes the INITIAL value of the fp32 stream).
// NCH > 0 (d == 512 * NCH: 4096 and 1024, the two widths of the model): the row is requested ONCE, all 2 * NCH float4 per lane
// before anything is consumed, and the three passes run on registers - the generic form below (NCH = 0) walks the row three times
// with one dependent 32-byte request per lane and pass in flight.
template <int DT, bool LN, bool Y32 = false, int NCH = 0>
__global__ __launch_bounds__(256) void norm_fwd_f32in_k(const float* __restrict__ x, const uint16_t* __restrict__ w, const uint16_t* __restrict__ b,
                                                        uint16_t* __restrict__ y, uint16_t* __restrict__ x16, int rows, int d, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * ROWS_PER_BLOCK + wave;
  if (row >= rows) return;
  const float4* xr = (const float4*)(x + (int64_t)row * d);
  const uint4* wr = (const uint4*)w;
  const uint4* br = (const uint4*)b;
  uint4* yr = (uint4*)(y + (int64_t)row * d);
  uint4* cr = x16 ? (uint4*)(x16 + (int64_t)row * d) : nullptr;
  const int nch = d >> 3;
  if constexpr (NCH > 0) {
    float4 xa[NCH], xe[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      xa[j] = xr[2 * c];
      xe[j] = xr[2 * c + 1];
    }
    uint4 wq[NCH], bq[LN ? NCH : 1];  // requested now as well: a load in the last pass would wait behind that pass's stores (one vmcnt)
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      wq[j] = wr[lane + 64 * j];
      if constexpr (LN) bq[j] = br[lane + 64 * j];
    }
    float mu = 0.f;
    if constexpr (LN) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const float4 a = xa[j], e = xe[j];
        s += a.x; s += a.y; s += a.z; s += a.w; s += e.x; s += e.y; s += e.z; s += e.w;
      }
      mu = wave_sum(s) / (float)d;
    }
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const float4 a = xa[j], e = xe[j];
      const float f[8] = {a.x, a.y, a.z, a.w, e.x, e.y, e.z, e.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) ss += (f[i] - mu) * (f[i] - mu);
    }
    const float r = rsqrtf(wave_sum(ss) / (float)d + eps);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      const float4 a = xa[j], e = xe[j];
      float f[8] = {a.x, a.y, a.z, a.w, e.x, e.y, e.z, e.w}, g[8];
      if (cr) cr[c] = pack8<DT>(f);
      unpack8<DT>(wq[j], g);
      if constexpr (LN) {
        float h[8];
        unpack8<DT>(bq[j], h);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (f[i] - mu) * r * g[i] + h[i];
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = f[i] * r * g[i];
      }
      if constexpr (Y32) {
        float4* y4 = (float4*)((float*)y + (int64_t)row * d);
        y4[2 * c] = make_float4(f[0], f[1], f[2], f[3]);
        y4[2 * c + 1] = make_float4(f[4], f[5], f[6], f[7]);
      } else {
        yr[c] = pack8<DT>(f);
      }
    }
    return;
  }
  float mu = 0.f;
  if constexpr (LN) {
    float s = 0.f;
    for (int c = lane; c < nch; c += 64) {
      const float4 a = xr[2 * c], e = xr[2 * c + 1];
      s += a.x; s += a.y; s += a.z; s += a.w; s += e.x; s += e.y; s += e.z; s += e.w;
    }
    mu = wave_sum(s) / (float)d;
  }
  float ss = 0.f;
  for (int c = lane; c < nch; c += 64) {
    const float4 a = xr[2 * c], e = xr[2 * c + 1];
    const float f[8] = {a.x, a.y, a.z, a.w, e.x, e.y, e.z, e.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += (f[i] - mu) * (f[i] - mu);
  }
  const float r = rsqrtf(wave_sum(ss) / (float)d + eps);
  for (int c = lane; c < nch; c += 64) {
    const float4 a = xr[2 * c], e = xr[2 * c + 1];
    float f[8] = {a.x, a.y, a.z, a.w, e.x, e.y, e.z, e.w}, g[8];
    if (cr) cr[c] = pack8<DT>(f);
    unpack8<DT>(wr[c], g);
    if constexpr (LN) {
      float h[8];
      unpack8<DT>(br[c], h);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = (f[i] - mu) * r * g[i] + h[i];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = f[i] * r * g[i];
    }
    if constexpr (Y32) {
      float4* y4 = (float4*)((float*)y + (int64_t)row * d);
      y4[2 * c] = make_float4(f[0], f[1], f[2], f[3]);
      y4[2 * c + 1] = make_float4(f[4], f[5], f[6], f[7]);
    } else {
      yr[c] = pack8<DT>(f);
    }
  }
}

// Backward.  LN = false: RMSNorm, LN = true: LayerNorm.  NCH = chunks (of 8 elements) per lane.
template <int DT, bool LN, int NCH>
__global__ __launch_bounds__(256) void norm_bwd_k(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                  const uint16_t* __restrict__ dy, uint16_t* __restrict__ dx,
                                                  float* __restrict__ dw_partial, float* __restrict__ db_partial,
                                                  int rows, int d, float eps, int rows_per_block, int accumulate_dx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = d >> 3;
  const uint4* wr = (const uint4*)w;
  float dwacc[NCH][8];
  float dbacc[LN ? NCH : 1][8];
#pragma unroll
  for (int j = 0; j < NCH; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      dwacc[j][i] = 0.f;
      if (LN) dbacc[j][i] = 0.f;
    }
  const int r_begin = blockIdx.x * rows_per_block;
  const int r_end = min(rows, r_begin + rows_per_block);
  const float invd = 1.0f / (float)d;
  // the weight row is loop-invariant: keep it packed in registers
  uint4 wraw[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = lane + 64 * j;
    wraw[j] = c < nch ? wr[c] : make_uint4(0, 0, 0, 0);
  }
  for (int row = r_begin + wave; row < r_end; row += ROWS_PER_BLOCK) {
    const uint4* xr = (const uint4*)(x + (int64_t)row * d);
    const uint4* gr = (const uint4*)(dy + (int64_t)row * d);
    uint4* dxr = (uint4*)(dx + (int64_t)row * d);
    // every load of the row (x, dy and, when accumulating, the old dx) is issued before anything is consumed; the
    // values stay PACKED in registers (3 x NCH uint4) and are unpacked on the fly in each pass - three times the
    // bytes in flight per wave of the unpack-first form, which ran at 3.7 TB/s
    uint4 xraw[NCH], graw[NCH], praw[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      if (c < nch) {
        xraw[j] = xr[c];
        graw[j] = gr[c];
        if (accumulate_dx) praw[j] = dxr[c];
      }
    }
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      if (c < nch) {
        float xs[8];
        unpack8<DT>(xraw[j], xs);
#pragma unroll
        for (int i = 0; i < 8; ++i) s1 += LN ? xs[i] : xs[i] * xs[i];
      }
    }
    float mu = 0.f, r;
    if (LN) {
      mu = wave_sum(s1) * invd;
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int c = lane + 64 * j;
        if (c < nch) {
          float xs[8];
          unpack8<DT>(xraw[j], xs);
#pragma unroll
          for (int i = 0; i < 8; ++i) ss += (xs[i] - mu) * (xs[i] - mu);
        }
      }
      r = rsqrtf(wave_sum(ss) * invd + eps);
    } else {
      r = rsqrtf(wave_sum(s1) * invd + eps);
    }
    // (opaque touches: without them hipcc keeps the UNPACKED fp32 copies alive across the passes - 405 VGPRs, one wave
    // per SIMD - instead of re-unpacking the packed registers)
#pragma unroll
    for (int j = 0; j < NCH; ++j) { TOUCH4(xraw[j]); TOUCH4(graw[j]); }
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      if (c < nch) {
        float xs[8], dyf[8], wf[8];
        unpack8<DT>(xraw[j], xs);
        unpack8<DT>(graw[j], dyf);
        unpack8<DT>(wraw[j], wf);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xh = (xs[i] - mu) * r;  // xhat (RMS: mu = 0)
          const float gsv = wf[i] * dyf[i];
          dwacc[j][i] += dyf[i] * xh;
          if (LN) dbacc[j][i] += dyf[i];
          sg += gsv;
          sgx += gsv * xh;
        }
      }
    }
    sgx = wave_sum(sgx) * invd;
    if (LN) sg = wave_sum(sg) * invd; else sg = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) { TOUCH4(xraw[j]); TOUCH4(graw[j]); TOUCH4(wraw[j]); }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      if (c < nch) {
        float xs[8], dyf[8], wf[8], o[8];
        unpack8<DT>(xraw[j], xs);
        unpack8<DT>(graw[j], dyf);
        unpack8<DT>(wraw[j], wf);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xh = (xs[i] - mu) * r;
          o[i] = r * (wf[i] * dyf[i] - sg - xh * sgx);
        }
        if (accumulate_dx) {
          float p[8];
          unpack8<DT>(praw[j], p);
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] += p[i];
        }
        dxr[c] = pack8<DT>(o);
      }
    }
  }
  // combine the 4 waves' dw/db accumulators through LDS, write this block's partial row
  float* red = (float*)smem;  // [4][d]
  const int npass = LN ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      if (c < nch) {
#pragma unroll
        for (int i = 0; i < 8; ++i) red[wave * d + c * 8 + i] = (pass == 0) ? dwacc[j][i] : dbacc[LN ? j : 0][i];
      }
    }
    __syncthreads();
    float* outp = (pass == 0 ? dw_partial : db_partial) + (int64_t)blockIdx.x * d;
    for (int e = threadIdx.x; e < d; e += 256) outp[e] = red[e] + red[d + e] + red[2 * d + e] + red[3 * d + e];
  }
}

template <int DT>
__global__ __launch_bounds__(1024) void reduce_partials_k(const float* __restrict__ partial, int nblk, int d,
                                                          void* __restrict__ out, int out_dt, int accumulate) {
  // 32 columns per block; 32 row groups split the partial rows (fixed order and tree: deterministic).  With up to
  // 1024 partial rows this is 32 independent loads per thread, 4 in flight, instead of 256 dependent ones.
  __shared__ float red[32][33];
  const int col = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + col;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < d) {
    int b = g;
    for (; b + 96 < nblk; b += 128) {
      s0 += partial[(int64_t)b * d + e];
      s1 += partial[(int64_t)(b + 32) * d + e];
      s2 += partial[(int64_t)(b + 64) * d + e];
      s3 += partial[(int64_t)(b + 96) * d + e];
    }
    for (; b < nblk; b += 32) s0 += partial[(int64_t)b * d + e];
  }
  red[g][col] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g != 0 || e >= d) return;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += red[i][col];
  if (out_dt == MH_F32) {
    float* o = (float*)out;
    o[e] = accumulate ? o[e] + s : s;
  } else {
    uint16_t* o = (uint16_t*)out;
    if (accumulate) s += ld16<DT>(o[e]);
    o[e] = (uint16_t)st16<DT>(s);
  }
}

// per-row-block column sums of x[rows, d] (ld = ldx): partial[blk, d]
template <int DT>
__global__ __launch_bounds__(256) void colsum_partial_k(const uint16_t* __restrict__ x, int64_t ldx,
                                                        float* __restrict__ partial, int rows, int d, int rows_per_block) {
  const int c = blockIdx.y * 256 + threadIdx.x;  // column
  if (c >= d) return;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += ld16<DT>(x[(int64_t)r * ldx + c]);
  partial[(int64_t)blockIdx.x * d + c] = s;
}

// the same for 16-byte-aligned rows with d % 8 == 0: a thread owns 8 consecutive columns (one 16-byte load per row, four rows in flight),
// 32 such threads cover 256 columns and the block's 8 thread rows take every eighth row; fixed-order sum through LDS.  (The 2-byte-per-
// lane form above streamed at 2.2 TB/s: 5 ms of the cfg-3 step in the tower's bias gradients.)
template <int DT>
__global__ __launch_bounds__(256) void colsum_partial_v8_k(const uint16_t* __restrict__ x, int64_t ldx, float* __restrict__ partial, int rows,
                                                           int d, int rows_per_block) {
  __shared__ float red[8][256];
  const int oc = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c0 = blockIdx.y * 256 + oc * 8;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = 0.f;
  if (c0 < d) {
    const uint16_t* xp = x + c0;
    int r = r0 + rl;
    for (; r + 24 < r1; r += 32) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *(const uint4*)(xp + (int64_t)(r + 8 * u) * ldx);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float f[8];
        unpack8<DT>(v[u], f);
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] += f[i];
      }
    }
    for (; r < r1; r += 8) {
      float f[8];
      unpack8<DT>(*(const uint4*)(xp + (int64_t)r * ldx), f);
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] += f[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[rl][oc * 8 + i] = s[i];
  __syncthreads();
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c < d) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) a += red[k][threadIdx.x];
    partial[(int64_t)blockIdx.x * d + c] = a;
  }
}

inline int partials_for(int rows) {
  int n = (rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  return n < MAX_PARTIALS ? n : MAX_PARTIALS;
}

template <int DT, bool LN>
int launch_norm_bwd(const void* x, const void* w, const void* dy, void* dx, float* dwp, float* dbp, int rows, int d,
                    float eps, int acc, hipStream_t st) {
  const int nblk = partials_for(rows);
  const int rpb = (rows + nblk - 1) / nblk;
  const int nch = d >> 3;
  const int per_lane = (nch + 63) / 64;
  const size_t lds = (size_t)4 * d * sizeof(float);
#define LAUNCH(N)                                                                                                   \
  hipLaunchKernelGGL((norm_bwd_k<DT, LN, N>), dim3(nblk), dim3(256), lds, st, (const uint16_t*)x, (const uint16_t*)w, \
                     (const uint16_t*)dy, (uint16_t*)dx, dwp, dbp, rows, d, eps, rpb, acc)
  if (per_lane <= 1) LAUNCH(1);
  else if (per_lane <= 2) LAUNCH(2);
  else if (per_lane <= 4) LAUNCH(4);
  else if (per_lane <= 8) LAUNCH(8);
  else return MH_ERR_SHAPE;
#undef LAUNCH
  MH_LAUNCH_CHECK();
}

}  // namespace

extern "C" int mh_norm_bwd_partials(int rows) { return partials_for(rows); }

extern "C" int mh_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int rows, int d, float eps, int dt,
                              void* stream) {
  if (!x || !w || !y || rows <= 0 || d <= 0 || (d & 7)) return MH_ERR_ARG;
  if (rows <= 64 && d <= 8192 && aligned16(x) && aligned16(w) && aligned16(y)) {  // a few rows (decode): a block per row
    if (dt == MH_BF16)
      hipLaunchKernelGGL(rmsnorm_fwd_row_k<MH_BF16>, dim3(rows), dim3(256), 0, as_stream(stream), (const uint16_t*)x, (const uint16_t*)w,
                         (uint16_t*)y, rstd, d, eps);
    else if (dt == MH_F16)
      hipLaunchKernelGGL(rmsnorm_fwd_row_k<MH_F16>, dim3(rows), dim3(256), 0, as_stream(stream), (const uint16_t*)x, (const uint16_t*)w,
                         (uint16_t*)y, rstd, d, eps);
    else return MH_ERR_DTYPE;
    MH_LAUNCH_CHECK();
  }
  const int grid = (rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  if (dt == MH_BF16)
    hipLaunchKernelGGL(rmsnorm_fwd_k<MH_BF16>, dim3(grid), dim3(256), 0, as_stream(stream), (const uint16_t*)x,
                       (const uint16_t*)w, (uint16_t*)y, rstd, rows, d, eps);
  else if (dt == MH_F16)
    hipLaunchKernelGGL(rmsnorm_fwd_k<MH_F16>, dim3(grid), dim3(256), 0, as_stream(stream), (const uint16_t*)x,
                       (const uint16_t*)w, (uint16_t*)y, rstd, rows, d, eps);
  else return MH_ERR_DTYPE;
  MH_LAUNCH_CHECK();
}
extern "C" int mh_rmsnorm_fwd_q8(const void* x, const void* w, void* y, void* q, float* scales, int rows, int d, float eps, int dt, void* stream) {
  if (!x || !w || !y || !q || !scales || rows <= 0 || d <= 0 || (d & 7) || !aligned16(x) || !aligned16(w) || !aligned16(y) || (((uintptr_t)q) & 7u)) return MH_ERR_ARG;
  const dim3 grid((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), block(256);
  if (dt != MH_BF16 && dt != MH_F16) return MH_ERR_DTYPE;
#define Q8_GO(DT_, NCH_)                                                                                                                \
  hipLaunchKernelGGL((rmsnorm_fwd_q8_k<DT_, NCH_>), grid, block, 0, as_stream(stream), (const uint16_t*)x, (const uint16_t*)w, (uint16_t*)y, \
                     (uint8_t*)q, scales, rows, d, eps)
  if (dt == MH_BF16) { if (d == 4096) Q8_GO(MH_BF16, 8); else Q8_GO(MH_BF16, 0); }
  else { if (d == 4096) Q8_GO(MH_F16, 8); else Q8_GO(MH_F16, 0); }
#undef Q8_GO
  MH_LAUNCH_CHECK();
}


extern "C" int mh_layernorm_fwd(const void* x, const void* w, const void* b, void* y, int rows, int d, float eps,
                                int dt, void* stream) {
  if (!x || !w || !b || !y || rows <= 0 || d <= 0 || (d & 7)) return MH_ERR_ARG;
  const int grid = (rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  if (dt == MH_BF16)
    hipLaunchKernelGGL(layernorm_fwd_k<MH_BF16>, dim3(grid), dim3(256), 0, as_stream(stream), (const uint16_t*)x,
                       (const uint16_t*)w, (const uint16_t*)b, (uint16_t*)y, rows, d, eps);
  else if (dt == MH_F16)
    hipLaunchKernelGGL(layernorm_fwd_k<MH_F16>, dim3(grid), dim3(256), 0, as_stream(stream), (const uint16_t*)x,
                       (const uint16_t*)w, (const uint16_t*)b, (uint16_t*)y, rows, d, eps);
  else return MH_ERR_DTYPE;
  MH_LAUNCH_CHECK();
}

extern "C" int mh_norm_fwd_f32in(const float* x, const void* w, const void* b, void* y, void* x16, int rows, int d, float eps, int dt,
                                 void* stream) {
  if (!x || !w || !y || rows <= 0 || d <= 0 || (d & 7)) return MH_ERR_ARG;
  if (!aligned16(x) || !aligned16(w) || !aligned16(y) || (b && !aligned16(b)) || (x16 && !aligned16(x16))) return MH_ERR_ARG;
  if (dt != MH_BF16 && dt != MH_F16) return MH_ERR_DTYPE;
  const int grid = (rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
#define NF32_GO_(DT_, LN_, NCH_)                                                                                                \
  hipLaunchKernelGGL((norm_fwd_f32in_k<DT_, LN_, false, NCH_>), dim3(grid), dim3(256), 0, as_stream(stream), x, (const uint16_t*)w,   \
                     (const uint16_t*)b, (uint16_t*)y, (uint16_t*)x16, rows, d, eps)
#define NF32_GO(DT_, LN_)                                                                                                       \
  do { if (d == 4096) NF32_GO_(DT_, LN_, 8); else if (d == 1024) NF32_GO_(DT_, LN_, 2); else NF32_GO_(DT_, LN_, 0); } while (0)
  if (dt == MH_BF16) { if (b) NF32_GO(MH_BF16, true); else NF32_GO(MH_BF16, false); }
  else { if (b) NF32_GO(MH_F16, true); else NF32_GO(MH_F16, false); }
#undef NF32_GO
#undef NF32_GO_
  MH_LAUNCH_CHECK();
}

// LayerNorm of an fp32 tensor INTO an fp32 tensor (+ the 16-bit copy of the input the backward keeps): the CLIP tower's pre_layrnorm when its
// residual stream is fp32 (engine.fp32_residual) - the stream then starts from unrounded values.
extern "C" int mh_layernorm_f32_to_f32(const float* x, const void* w, const void* b, float* y, void* x16, int rows, int d, float eps, int dt, void* stream) {
  if (!x || !w || !b || !y || rows <= 0 || d <= 0 || (d & 7)) return MH_ERR_ARG;
  if (!aligned16(x) || !aligned16(w) || !aligned16(y) || !aligned16(b) || (x16 && !aligned16(x16))) return MH_ERR_ARG;
  const int grid = (rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  if (dt == MH_BF16)
    hipLaunchKernelGGL((norm_fwd_f32in_k<MH_BF16, true, true>), dim3(grid), dim3(256), 0, as_stream(stream), x, (const uint16_t*)w, (const uint16_t*)b,
                       (uint16_t*)y, (uint16_t*)x16, rows, d, eps);
  else if (dt == MH_F16)
    hipLaunchKernelGGL((norm_fwd_f32in_k<MH_F16, true, true>), dim3(grid), dim3(256), 0, as_stream(stream), x, (const uint16_t*)w, (const uint16_t*)b,
                       (uint16_t*)y, (uint16_t*)x16, rows, d, eps);
  else return MH_ERR_DTYPE;
  MH_LAUNCH_CHECK();
}

extern "C" int mh_rmsnorm_bwd(const void* x, const void* w, const void* dy, void* dx, float* dw_partial, int rows,
                              int d, float eps, int dt, int accumulate_dx, void* stream) {
  if (!x || !w || !dy || !dx || !dw_partial || rows <= 0 || (d & 7)) return MH_ERR_ARG;
  if (dt == MH_BF16) return launch_norm_bwd<MH_BF16, false>(x, w, dy, dx, dw_partial, nullptr, rows, d, eps, accumulate_dx, as_stream(stream));
  if (dt == MH_F16) return launch_norm_bwd<MH_F16, false>(x, w, dy, dx, dw_partial, nullptr, rows, d, eps, accumulate_dx, as_stream(stream));
  return MH_ERR_DTYPE;
}

extern "C" int mh_layernorm_bwd(const void* x, const void* w, const void* dy, void* dx, float* dw_partial,
                                float* db_partial, int rows, int d, float eps, int dt, int accumulate_dx, void* stream) {
  if (!x || !w || !dy || !dx || !dw_partial || !db_partial || rows <= 0 || (d & 7)) return MH_ERR_ARG;
  if (dt == MH_BF16) return launch_norm_bwd<MH_BF16, true>(x, w, dy, dx, dw_partial, db_partial, rows, d, eps, accumulate_dx, as_stream(stream));
  if (dt == MH_F16) return launch_norm_bwd<MH_F16, true>(x, w, dy, dx, dw_partial, db_partial, rows, d, eps, accumulate_dx, as_stream(stream));
  return MH_ERR_DTYPE;
}

extern "C" int mh_reduce_partials(const float* partial, int nblk, int d, void* out, int dt, int accumulate, void* stream) {
  if (!partial || !out || nblk <= 0 || d <= 0) return MH_ERR_ARG;
  const int grid = (d + 31) / 32;
  if (dt == MH_F16)
    hipLaunchKernelGGL(reduce_partials_k<MH_F16>, dim3(grid), dim3(1024), 0, as_stream(stream), partial, nblk, d, out, dt, accumulate);
  else
    hipLaunchKernelGGL(reduce_partials_k<MH_BF16>, dim3(grid), dim3(1024), 0, as_stream(stream), partial, nblk, d, out, dt, accumulate);
  MH_LAUNCH_CHECK();
}

extern "C" int mh_colsum_partial(const void* x, int64_t ldx, float* partial, int rows, int d, int dt, void* stream) {
  if (!x || !partial || rows <= 0 || d <= 0) return MH_ERR_ARG;
  const int nblk = partials_for(rows);
  const int rpb = (rows + nblk - 1) / nblk;
  dim3 grid(nblk, (d + 255) / 256);
  if ((d & 7) == 0 && (ldx & 7) == 0 && aligned16(x) && (dt == MH_BF16 || dt == MH_F16)) {
    if (dt == MH_BF16)
      hipLaunchKernelGGL(colsum_partial_v8_k<MH_BF16>, grid, dim3(256), 0, as_stream(stream), (const uint16_t*)x, ldx, partial, rows, d, rpb);
    else
      hipLaunchKernelGGL(colsum_partial_v8_k<MH_F16>, grid, dim3(256), 0, as_stream(stream), (const uint16_t*)x, ldx, partial, rows, d, rpb);
    MH_LAUNCH_CHECK();
  }
  if (dt == MH_BF16)
    hipLaunchKernelGGL(colsum_partial_k<MH_BF16>, grid, dim3(256), 0, as_stream(stream), (const uint16_t*)x, ldx, partial, rows, d, rpb);
  else if (dt == MH_F16)
    hipLaunchKernelGGL(colsum_partial_k<MH_F16>, grid, dim3(256), 0, as_stream(stream), (const uint16_t*)x, ldx, partial, rows, d, rpb);
  else return MH_ERR_DTYPE;
  MH_LAUNCH_CHECK();
}
