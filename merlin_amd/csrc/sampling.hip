// Token selection for generate() (eval_mmvet.py:101-120 calls model.generate(do_sample=True, temperature=0.2, ...) or
// num_beams=5): the logits processors + samplers HF's GenerationMixin applies to the last-position logits, as HBM/L2-bound
// row kernels (one 256-thread block per sequence; a row of 32003 fp32 logits is 128 KB and is re-read from L2).
//
//   mh_select_tokens   argmax (do_sample=0) or temperature -> top-k -> top-p -> multinomial (inverse CDF in index order; the
//                      uniform comes from a counter-based generator keyed on (seed, step, row): no host RNG traffic, replayable)
//   mh_log_softmax_rows  fp32 log-probabilities for beam search
//   mh_gather_rows2d   dst[i, :cols] = src[idx[i], :cols], zeros where idx[i] < 0  (KV-cache reorder by beam index, batch expansion,
//                      unpad / pad of the key-padding attention branch)
#include "mh_common.h"

namespace {

// order-preserving map float -> uint32 (larger float <-> larger key); -0.0 sorts below +0.0, NaN keys are never produced
// because the callers skip non-finite comparisons by construction (logits are finite)
__device__ __forceinline__ uint32_t fkey(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

struct BlockRed {
  float* fs;
  int* is;
  __device__ __forceinline__ float sum(float v) {  // fixed tree order: deterministic
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) fs[threadIdx.x >> 6] = v;
    __syncthreads();
    return (fs[0] + fs[1]) + (fs[2] + fs[3]);
  }
  __device__ __forceinline__ float max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) fs[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(fs[0], fs[1]), fmaxf(fs[2], fs[3]));
  }
  __device__ __forceinline__ int isum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) is[threadIdx.x >> 6] = v;
    __syncthreads();
    return is[0] + is[1] + is[2] + is[3];
  }
};

// One block per row.  z_i = logits_i * inv_t.
//   do_sample == 0: out = lowest index of the maximum.
//   else keep = { z_i >= t_k } (t_k = k-th largest value; ties kept, like HF's TopKLogitsWarper `scores < kth`), then
//        keep &= { mass of strictly larger kept tokens < top_p } (TopPLogitsWarper: tokens whose ascending cumulative
//        probability is <= 1 - top_p are removed, at least one is kept), then multinomial over softmax(z | keep).
__global__ __launch_bounds__(256) void select_tokens_k(const float* __restrict__ logits, int64_t ldl, int V, int do_sample,
                                                       float inv_t, int top_k, float top_p, uint64_t seed, int64_t step,
                                                       int64_t* __restrict__ out, float* __restrict__ out_u) {
  __shared__ float fs[4];
  __shared__ int is[4];
  __shared__ float chunk_sum[256];
  __shared__ int sh_idx;
  BlockRed red{fs, is};
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* x = logits + (int64_t)row * ldl;
  // contiguous chunk per thread: the CDF runs in index order
  const int C = (V + 255) / 256, i0 = tid * C, i1 = min(V, i0 + C);

  float m = -INFINITY;
  int am = 0x7fffffff;
  for (int i = i0; i < i1; ++i) {
    const float z = x[i] * inv_t;
    if (z > m) { m = z; am = i; }
  }
  const float gm = red.max(m);
  if (!do_sample) {
    int cand = (m == gm) ? am : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    __syncthreads();
    if ((tid & 63) == 0) is[tid >> 6] = cand;
    __syncthreads();
    if (tid == 0) out[row] = min(min(is[0], is[1]), min(is[2], is[3]));
    return;
  }
  // ---- top-k threshold: the largest key K with count{key >= K} >= k  (bisection over the 32-bit ordered key space)
  uint32_t kkey = 0;
  if (top_k > 0 && top_k < V) {
    uint32_t lo = 0, hi = 0xffffffffu;  // invariant: count{>= lo} >= k
    while (lo < hi) {
      const uint32_t mid = lo + (uint32_t)(((uint64_t)hi - lo + 1) >> 1);
      int c = 0;
      for (int i = i0; i < i1; ++i) c += fkey(x[i] * inv_t) >= mid;
      if (red.isum(c) >= top_k) lo = mid; else hi = mid - 1;
    }
    kkey = lo;
  }
  // ---- top-p threshold: the smallest key K (>= kkey) with mass{key > K | key >= kkey} < top_p * total
  float total = 0.f;
  for (int i = i0; i < i1; ++i) {
    const float z = x[i] * inv_t;
    if (fkey(z) >= kkey) total += __expf(z - gm);
  }
  total = red.sum(total);
  if (top_p < 1.0f) {
    const float lim = top_p * total;
    uint32_t lo = kkey, hi = fkey(gm);  // mass{> fkey(gm)} = 0 < lim: hi always satisfies the predicate
    while (lo < hi) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      float s = 0.f;
      for (int i = i0; i < i1; ++i) {
        const float z = x[i] * inv_t;
        if (fkey(z) > mid) s += __expf(z - gm);
      }
      if (red.sum(s) < lim) hi = mid; else lo = mid + 1;
    }
    kkey = lo;
    total = 0.f;
    for (int i = i0; i < i1; ++i) {
      const float z = x[i] * inv_t;
      if (fkey(z) >= kkey) total += __expf(z - gm);
    }
    total = red.sum(total);
  }
  // ---- multinomial: inverse CDF in index order
  float cs = 0.f;
  for (int i = i0; i < i1; ++i) {
    const float z = x[i] * inv_t;
    if (fkey(z) >= kkey) cs += __expf(z - gm);
  }
  chunk_sum[tid] = cs;
  if (tid == 0) sh_idx = -1;
  __syncthreads();
  const uint64_t r = splitmix64(seed ^ splitmix64((uint64_t)step * 0x100000001B3ull + (uint64_t)row));
  const float u = (float)(r >> 40) * (1.0f / 16777216.0f);  // [0, 1)
  if (tid == 0) {
    float tot = 0.f;
    for (int t = 0; t < 256; ++t) tot += chunk_sum[t];
    const float target = u * tot;
    float acc = 0.f;
    int t = 0, last = 0;
    for (; t < 256; ++t) {
      if (chunk_sum[t] > 0.f) last = t;
      if (acc + chunk_sum[t] > target && chunk_sum[t] > 0.f) break;
      acc += chunk_sum[t];
    }
    if (t == 256) { t = last; acc = tot - chunk_sum[last]; }  // rounding at the very end of the CDF
    const int j0 = t * C, j1 = min(V, j0 + C);
    int pick = -1, lastk = -1;
    for (int i = j0; i < j1; ++i) {
      const float z = x[i] * inv_t;
      if (fkey(z) < kkey) continue;
      lastk = i;
      acc += __expf(z - gm);
      if (acc > target) { pick = i; break; }
    }
    out[row] = pick >= 0 ? pick : lastk;
    if (out_u) out_u[row] = u;
  }
}

__global__ __launch_bounds__(256) void log_softmax_rows_k(const float* __restrict__ logits, int64_t ldl, int V, float* __restrict__ out,
                                                          int64_t ldo, const float* __restrict__ row_bias) {
  __shared__ float fs[4];
  __shared__ int is[4];
  BlockRed red{fs, is};
  const int row = blockIdx.x;
  const float* x = logits + (int64_t)row * ldl;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < V; i += 256) m = fmaxf(m, x[i]);
  m = red.max(m);
  float s = 0.f;
  for (int i = threadIdx.x; i < V; i += 256) s += expf(x[i] - m);
  s = red.sum(s);
  const float lse = m + logf(s) - (row_bias ? row_bias[row] : 0.f);
  for (int i = threadIdx.x; i < V; i += 256) out[(int64_t)row * ldo + i] = x[i] - lse;
}

// VT-sized vectors (16 bytes when every stride / width / pointer allows it, else 4 bytes)
template <typename VT>
__global__ __launch_bounds__(256) void gather_rows2d_k(const VT* __restrict__ src, int64_t lds_v, const int64_t* __restrict__ idx,
                                                       VT* __restrict__ dst, int64_t ldd_v, int rows, int64_t cols_v) {
  const int64_t total = (int64_t)rows * cols_v;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cols_v, c = i - r * cols_v;
    const int64_t sr = idx[r];  // < 0: a zero row (pad_input of the key-padding branch: rows no valid token maps to)
    VT val = VT{};
    if (sr >= 0) val = src[sr * lds_v + c];
    dst[r * ldd_v + c] = val;
  }
}

}  // namespace

extern "C" int mh_select_tokens(const float* logits, int64_t ldl, int rows, int V, int do_sample, float temperature, int top_k,
                                float top_p, uint64_t seed, int64_t step, int64_t* out, float* out_u, void* stream) {
  if (!logits || !out || rows <= 0 || V <= 0 || ldl < V) return MH_ERR_ARG;
  if (do_sample && (!(temperature > 0.f) || !(top_p > 0.f) || top_p > 1.0f || top_k < 0)) return MH_ERR_ARG;
  hipLaunchKernelGGL(select_tokens_k, dim3(rows), dim3(256), 0, as_stream(stream), logits, ldl, V, do_sample,
                     do_sample ? 1.0f / temperature : 1.0f, top_k, top_p, seed, step, out, out_u);
  MH_LAUNCH_CHECK();
}

extern "C" int mh_log_softmax_rows(const float* logits, int64_t ldl, int rows, int V, float* out, int64_t ldo, const float* row_bias,
                                   void* stream) {
  if (!logits || !out || rows <= 0 || V <= 0 || ldl < V || ldo < V) return MH_ERR_ARG;
  hipLaunchKernelGGL(log_softmax_rows_k, dim3(rows), dim3(256), 0, as_stream(stream), logits, ldl, V, out, ldo, row_bias);
  MH_LAUNCH_CHECK();
}

extern "C" int mh_gather_rows2d(const void* src, int64_t lds_bytes, const int64_t* idx, void* dst, int64_t ldd_bytes, int rows,
                                int64_t cols_bytes, void* stream) {
  if (!src || !idx || !dst || rows <= 0 || cols_bytes <= 0) return MH_ERR_ARG;
  const uint64_t all = (uint64_t)lds_bytes | (uint64_t)ldd_bytes | (uint64_t)cols_bytes | (uint64_t)(uintptr_t)src | (uint64_t)(uintptr_t)dst;
  if (all & 3) return MH_ERR_ARG;
  const int vb = (all & 15) ? 4 : 16;
  const int64_t nvec = (int64_t)rows * (cols_bytes / vb);
  int64_t b = (nvec + 255) / 256;
  const int grid = (int)(b < 4096 ? (b > 0 ? b : 1) : 4096);
  if (vb == 16)
    hipLaunchKernelGGL(gather_rows2d_k<uint4>, dim3(grid), dim3(256), 0, as_stream(stream), (const uint4*)src, lds_bytes >> 4, idx, (uint4*)dst,
                       ldd_bytes >> 4, rows, cols_bytes >> 4);
  else
    hipLaunchKernelGGL(gather_rows2d_k<uint32_t>, dim3(grid), dim3(256), 0, as_stream(stream), (const uint32_t*)src, lds_bytes >> 2, idx,
                       (uint32_t*)dst, ldd_bytes >> 2, rows, cols_bytes >> 2);
  MH_LAUNCH_CHECK();
}
