// Shifted cross-entropy (llama_mmgpt.py:92-100), embedding lookup + image-feature splice
// (base_mmgpt.py:99-160) and library-level queries.  HBM-bound / index kernels for gfx950.
#include "mh_common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Cross entropy.  One 256-thread block per token row; logits fp32, 16-byte loads.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void online_merge(float& m, float& s, float m2, float s2) {
  const float mn = fmaxf(m, m2);
  if (mn == -INFINITY) { m = mn; s = 0.f; return; }
  s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
  m = mn;
}

__device__ __forceinline__ int64_t shifted_label(const int64_t* labels, int64_t t, int S) {
  const int s = (int)(t % S);
  if (s == S - 1) return -100;
  return labels[t + 1];
}

__global__ __launch_bounds__(256) void ce_fwd_k(const float* __restrict__ logits, int64_t ldl, const int64_t* __restrict__ labels,
                                                float* __restrict__ row_loss, float* __restrict__ lse_out, int S, int V) {
  __shared__ float sm[4], ss[4];
  const int64_t t = blockIdx.x;
  const float* row = logits + t * ldl;
  float m = -INFINITY, s = 0.f;
  const int nv4 = ((ldl & 3) == 0) ? (V >> 2) : 0;
  for (int i = threadIdx.x; i < nv4; i += 256) {
    const float4 x = ((const float4*)row)[i];
    const float mx = fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w));
    const float mn = fmaxf(m, mx);
    s = s * __expf(m - mn) + __expf(x.x - mn) + __expf(x.y - mn) + __expf(x.z - mn) + __expf(x.w - mn);
    m = mn;
  }
  for (int i = nv4 * 4 + threadIdx.x; i < V; i += 256) {
    const float x = row[i];
    const float mn = fmaxf(m, x);
    s = s * __expf(m - mn) + __expf(x - mn);
    m = mn;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    online_merge(m, s, m2, s2);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sm[wave] = m; ss[wave] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float M = sm[0], Sx = ss[0];
    for (int w = 1; w < 4; ++w) online_merge(M, Sx, sm[w], ss[w]);
    const float lse = M + __logf(Sx);
    lse_out[t] = lse;
    const int64_t lab = shifted_label(labels, t, S);
    row_loss[t] = (lab >= 0 && lab < V) ? lse - row[lab] : 0.f;
  }
}

// out2[0] = sum(row_loss), out2[1] = #scored rows.  Single block, fixed order: deterministic.
__global__ __launch_bounds__(1024) void ce_reduce_k(const float* __restrict__ row_loss, const int64_t* __restrict__ labels,
                                                    float* __restrict__ out2, int64_t T, int S, int V) {
  __shared__ float sl[16], sc[16];
  float a = 0.f, c = 0.f;
  for (int64_t t = threadIdx.x; t < T; t += 1024) {
    const int64_t lab = shifted_label(labels, t, S);
    if (lab >= 0 && lab < V) { a += row_loss[t]; c += 1.f; }
  }
  a = wave_sum(a);
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) { sl[threadIdx.x >> 6] = a; sc[threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float A = 0.f, C = 0.f;
    for (int w = 0; w < 16; ++w) { A += sl[w]; C += sc[w]; }
    out2[0] = A;
    out2[1] = C;
    out2[2] = A / C;  // mean CE; NaN when no label is scored, like torch's CrossEntropyLoss
  }
}

template <int DT>
__global__ __launch_bounds__(256) void ce_bwd_k(const float* __restrict__ logits, int64_t ldl, const int64_t* __restrict__ labels,
                                                const float* __restrict__ lse, const float* __restrict__ out2,
                                                uint16_t* __restrict__ dlogits, int64_t lddl, int S, int V, int Vpad, float gscale,
                                                const int64_t* __restrict__ rows) {
  // rows != null: COMPACT form - output row r is the gradient of logits row rows[r] (rows[r] < 0: a zero row)
  const int64_t r = blockIdx.x;
  const int64_t tt = rows ? rows[r] : r;
  const int64_t t = tt < 0 ? 0 : tt;
  const int64_t lab = tt < 0 ? -100 : shifted_label(labels, t, S);
  const bool scored = (lab >= 0 && lab < V);
  const float cnt = out2[1];
  const float g = (scored && cnt > 0.f) ? gscale / cnt : 0.f;
  const float l = lse[t];
  const float* row = logits + t * ldl;
  uint16_t* drow = dlogits + r * lddl;
  for (int v = threadIdx.x; v < Vpad; v += 256) {
    float d = 0.f;
    if (scored && v < V) d = g * (__expf(row[v] - l) - (v == lab ? 1.f : 0.f));
    drow[v] = (uint16_t)st16<DT>(d);
  }
}

// ---------------------------------------------------------------------------------------------
// Splice index: one block per sample.
// ---------------------------------------------------------------------------------------------
constexpr int MAX_STARTS = 1024;

__global__ __launch_bounds__(256) void splice_index_k(const int64_t* __restrict__ ids, const int32_t* __restrict__ img_offset,
                                                      int32_t* __restrict__ src, int32_t* __restrict__ err, int S, int P,
                                                      int64_t im_patch, int64_t im_start, int64_t im_end, int rows_per_img, int row0) {
  __shared__ int cnt_start[256], cnt_end[256], cnt_patch[256];
  __shared__ int start_pos[MAX_STARTS];
  __shared__ int tot[3];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t* row = ids + (int64_t)b * S;
  int32_t* srow = src + (int64_t)b * S;
  const int chunk = (S + 255) / 256;
  const int p0 = tid * chunk, p1 = min(S, p0 + chunk);
  int ns = 0, ne = 0, np = 0;
  for (int p = p0; p < p1; ++p) {
    const int64_t v = row[p];
    ns += (v == im_start);
    ne += (v == im_end);
    np += (v == im_patch);
    srow[p] = -1;
  }
  cnt_start[tid] = ns; cnt_end[tid] = ne; cnt_patch[tid] = np;
  __syncthreads();
  if (tid == 0) {
    int a = 0, e = 0, c = 0;
    for (int i = 0; i < 256; ++i) {
      const int x = cnt_start[i];
      cnt_start[i] = a;  // exclusive prefix
      a += x; e += cnt_end[i]; c += cnt_patch[i];
    }
    tot[0] = a; tot[1] = e; tot[2] = c;
  }
  __syncthreads();
  if (tot[2] == 0) return;  // not a multimodal sample (base_mmgpt.py:109-113)
  if (tot[0] != tot[1]) {   // base_mmgpt.py:116-118
    if (tid == 0) { atomicExch(&err[0], 1); err[2] = b; err[3] = tot[0] - tot[1]; }
    return;
  }
  {
    int k = cnt_start[tid];
    for (int p = p0; p < p1; ++p)
      if (row[p] == im_start) { if (k < MAX_STARTS) start_pos[k] = p; ++k; }
  }
  __syncthreads();
  const int n_img = img_offset[b + 1] - img_offset[b];
  const int n_use = min(min(tot[0], n_img), MAX_STARTS);  // zip(): extra images / starts are ignored
  for (int k = 0; k < n_use; ++k) {
    const int p = start_pos[k];
    const int endp = p + P + 1;
    if (endp >= S || row[endp] != im_end) {  // base_mmgpt.py:125-126
      if (tid == 0) { atomicExch(&err[1], 1); err[2] = b; err[3] = p; }
      return;
    }
  }
  const int base = img_offset[b];
  for (int i = tid; i < n_use * P; i += 256) {
    const int k = i / P, j = i - k * P;
    srow[start_pos[k] + 1 + j] = (base + k) * rows_per_img + row0 + j;
  }
}

template <int DT>
__global__ __launch_bounds__(256) void embed_splice_fwd_k(const int64_t* __restrict__ ids, const int32_t* __restrict__ src,
                                                          const uint16_t* __restrict__ embed, const uint16_t* __restrict__ feats,
                                                          uint16_t* __restrict__ out, int64_t T, int d) {
  const int vpr = d >> 3;
  const int64_t total = T * vpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / vpr;
    const int v = (int)(i - t * vpr);
    const int s = src ? src[t] : -1;
    const uint16_t* from = (s >= 0) ? feats + (int64_t)s * d : embed + ids[t] * (int64_t)d;
    ((uint4*)(out + t * d))[v] = ((const uint4*)from)[v];
  }
}

// fp32 form (engine.fp32_residual: the decoder's residual stream starts from fp32 values): feats = the projector's fp32 output, embedding rows widened
template <int DT>
__global__ __launch_bounds__(256) void embed_splice_fwd_f32_k(const int64_t* __restrict__ ids, const int32_t* __restrict__ src,
                                                              const uint16_t* __restrict__ embed, const float* __restrict__ feats,
                                                              float* __restrict__ out, int64_t T, int d) {
  const int vpr = d >> 3;
  const int64_t total = T * vpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / vpr;
    const int v = (int)(i - t * vpr);
    const int s = src ? src[t] : -1;
    float4 lo, hi;
    if (s >= 0) {
      const float4* f4 = (const float4*)(feats + (int64_t)s * d);
      lo = f4[2 * v]; hi = f4[2 * v + 1];
    } else {
      float f[8];
      unpack8<DT>(((const uint4*)(embed + ids[t] * (int64_t)d))[v], f);
      lo = make_float4(f[0], f[1], f[2], f[3]); hi = make_float4(f[4], f[5], f[6], f[7]);
    }
    ((float4*)(out + t * d))[2 * v] = lo;
    ((float4*)(out + t * d))[2 * v + 1] = hi;
  }
}

template <int DT>
__global__ __launch_bounds__(256) void embed_splice_bwd_k(const int64_t* __restrict__ ids, const int32_t* __restrict__ src,
                                                          const uint16_t* __restrict__ dout, uint16_t* __restrict__ dfeats,
                                                          float* __restrict__ dembed32, int64_t T, int d) {
  const int vpr = d >> 3;
  const int64_t total = T * vpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / vpr;
    const int v = (int)(i - t * vpr);
    const int s = src ? src[t] : -1;
    const uint4 g = ((const uint4*)(dout + t * d))[v];
    if (s >= 0) {
      if (dfeats) ((uint4*)(dfeats + (int64_t)s * d))[v] = g;
    } else if (dembed32) {
      float f[8];
      unpack8<DT>(g, f);
      float* dst = dembed32 + ids[t] * (int64_t)d + v * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) atomicAdd(dst + k, f[k]);
    }
  }
}

// lens[b] = 1 + last position whose mask byte is non-zero (right-padded key_padding_mask -> length)
__global__ __launch_bounds__(256) void mask_lens_k(const uint8_t* __restrict__ mask, int32_t* __restrict__ lens, int S) {
  __shared__ int best[4];
  const int b = blockIdx.x;
  int m = 0;
  for (int s = threadIdx.x; s < S; s += 256)
    if (mask[(int64_t)b * S + s]) m = max(m, s + 1);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) best[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) lens[b] = max(max(best[0], best[1]), max(best[2], best[3]));
}

// unpad_input / pad_input of the flash-attention patch (llama_flash_attn_monkey_patch.py:87-102, flash_attn.bert_padding) as two row
// tables per sample, built by one block with a block-wide exclusive scan over the mask (order-preserving, deterministic):
//   fwd[b*S + r] = flat row of the r-th valid position of sample b (r < count[b]), -1 beyond   (unpad: compact = gather(x, fwd))
//   inv[b*S + s] = b*S + rank of position s among the valid ones, -1 where the mask is 0        (pad:   x = gather(compact, inv))
// The valid tokens of a sample are compacted to the FRONT OF ITS OWN S-row slot, so the attention kernels run on them as a right-padded
// batch with lens = count (causal order among valid tokens = the reference's causal attention over the packed sequence).
__global__ __launch_bounds__(256) void mask_unpad_index_k(const uint8_t* __restrict__ mask, int64_t* __restrict__ fwd, int64_t* __restrict__ inv,
                                                          int32_t* __restrict__ count, int S) {
  __shared__ int part[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int per = (S + 255) / 256, s0 = tid * per, s1 = min(S, s0 + per);
  const uint8_t* m = mask + (int64_t)b * S;
  int c = 0;
  for (int s = s0; s < s1; ++s) c += m[s] ? 1 : 0;
  part[tid] = c;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {  // Hillis-Steele inclusive scan
    const int v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int r = part[tid] - c;  // exclusive prefix of this thread's chunk
  const int total = part[255];
  const int64_t base = (int64_t)b * S;
  for (int s = s0; s < s1; ++s) {
    if (m[s]) {
      fwd[base + r] = base + s;
      inv[base + s] = base + r;
      ++r;
    } else {
      inv[base + s] = -1;
    }
  }
  for (int s = total + tid; s < S; s += 256) fwd[base + s] = -1;
  if (tid == 0) count[b] = total;
}

// Input validation on the device (one block per sample, no host sync): the reference raises on these through torch itself
// (embedding index out of range, CE target out of bounds) and honours arbitrary masks: a mask that is not a right-padded prefix is
// FLAGGED here (err[8]) and the host then routes attention through the unpad / pad tables above instead of the lens-only fast path.
//   err[4] = 1: an input id outside [0, V)      (err[5] = flat position)
//   err[6] = 1: a label that is neither -100 nor in [0, V)   (err[7] = flat position)
//   err[8] = 1: attention_mask of sample err[9] is not a right-padded prefix (popcount != 1 + last set position)
//   err[10] = 1: some sample's mask has a zero (the batch carries padding; 0 = every sequence is S long: the callers then run the
//                no-lengths forms of the attention kernels - no masks in the tile loops, five-product backward)
__global__ __launch_bounds__(256) void check_inputs_k(const int64_t* __restrict__ ids, const int64_t* __restrict__ labels,
                                                      const uint8_t* __restrict__ mask, const int32_t* __restrict__ lens,
                                                      int32_t* __restrict__ err, int S, int V) {
  __shared__ int cnt[4];
  const int b = blockIdx.x;
  int c = 0;
  for (int s = threadIdx.x; s < S; s += 256) {
    const int64_t t = (int64_t)b * S + s;
    if (ids) {
      const int64_t id = ids[t];
      if (id < 0 || id >= V) { if (atomicExch(&err[4], 1) == 0) err[5] = (int)t; }
    }
    if (labels) {
      const int64_t l = labels[t];
      if (l != -100 && (l < 0 || l >= V)) { if (atomicExch(&err[6], 1) == 0) err[7] = (int)t; }
    }
    if (mask && mask[t]) ++c;
  }
  if (mask) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int n = cnt[0] + cnt[1] + cnt[2] + cnt[3];
      if (n != lens[b]) { if (atomicExch(&err[8], 1) == 0) err[9] = b; }
      if (n != S) err[10] = 1;
    }
  }
}

inline int grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 2048 ? (b > 0 ? b : 1) : 2048);
}

}  // namespace

extern "C" int mh_version(void) { return 100; }

extern "C" int mh_arch_ok(int dev) {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
  const char* a = p.gcnArchName;
  return (a[0] == 'g' && a[1] == 'f' && a[2] == 'x' && a[3] == '9' && a[4] == '5' && a[5] == '0') ? 1 : 0;
}

extern "C" const char* mh_strerror(int code) {
  switch (code) {
    case MH_OK: return "ok";
    case MH_ERR_ARG: return "bad argument (null pointer, shape or alignment)";
    case MH_ERR_DTYPE: return "unsupported dtype";
    case MH_ERR_ARCH: return "device is not gfx950";
    case MH_ERR_SHAPE: return "shape not supported by this kernel";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}

extern "C" int mh_ce_fwd(const float* logits, int64_t ldl, const int64_t* labels, float* row_loss, float* lse, float* out2,
                         int B, int S, int V, void* stream) {
  if (!logits || !labels || !row_loss || !lse || !out2 || B <= 0 || S <= 0 || V <= 0) return MH_ERR_ARG;
  const int64_t T = (int64_t)B * S;
  hipLaunchKernelGGL(ce_fwd_k, dim3((unsigned)T), dim3(256), 0, as_stream(stream), logits, ldl, labels, row_loss, lse, S, V);
  hipLaunchKernelGGL(ce_reduce_k, dim3(1), dim3(1024), 0, as_stream(stream), (const float*)row_loss, labels, out2, T, S, V);
  MH_LAUNCH_CHECK();
}

static int ce_bwd_launch(const float* logits, int64_t ldl, const int64_t* labels, const float* lse, const float* out2, void* dlogits, int64_t lddl,
                         int64_t nrows, int S, int V, int Vpad, float gscale, const int64_t* rows, int dt, void* stream) {
  if (dt == MH_BF16)
    hipLaunchKernelGGL(ce_bwd_k<MH_BF16>, dim3((unsigned)nrows), dim3(256), 0, as_stream(stream), logits, ldl, labels, lse, out2, (uint16_t*)dlogits, lddl, S, V, Vpad, gscale, rows);
  else if (dt == MH_F16)
    hipLaunchKernelGGL(ce_bwd_k<MH_F16>, dim3((unsigned)nrows), dim3(256), 0, as_stream(stream), logits, ldl, labels, lse, out2, (uint16_t*)dlogits, lddl, S, V, Vpad, gscale, rows);
  else return MH_ERR_DTYPE;
  MH_LAUNCH_CHECK();
}
extern "C" int mh_ce_bwd(const float* logits, int64_t ldl, const int64_t* labels, const float* lse, const float* out2,
                         void* dlogits, int64_t lddl, int B, int S, int V, int Vpad, float gscale, int dt, void* stream) {
  if (!logits || !labels || !lse || !out2 || !dlogits || Vpad < V || lddl < Vpad) return MH_ERR_ARG;
  return ce_bwd_launch(logits, ldl, labels, lse, out2, dlogits, lddl, (int64_t)B * S, S, V, Vpad, gscale, nullptr, dt, stream);
}
extern "C" int mh_ce_bwd_rows(const float* logits, int64_t ldl, const int64_t* labels, const float* lse, const float* out2, void* dlogits, int64_t lddl,
                              const int64_t* rows, int nrows, int S, int V, int Vpad, float gscale, int dt, void* stream) {
  if (!logits || !labels || !lse || !out2 || !dlogits || !rows || nrows <= 0 || Vpad < V || lddl < Vpad) return MH_ERR_ARG;
  return ce_bwd_launch(logits, ldl, labels, lse, out2, dlogits, lddl, nrows, S, V, Vpad, gscale, rows, dt, stream);
}

extern "C" int mh_splice_index(const int64_t* ids, const int32_t* img_offset, int32_t* src, int32_t* err, int B, int S, int P,
                               int64_t im_patch, int64_t im_start, int64_t im_end, int rows_per_img, int row0, void* stream) {
  if (!ids || !img_offset || !src || !err || B <= 0 || S <= 0 || P <= 0 || rows_per_img < row0 + P) return MH_ERR_ARG;
  hipLaunchKernelGGL(splice_index_k, dim3(B), dim3(256), 0, as_stream(stream), ids, img_offset, src, err, S, P, im_patch, im_start, im_end, rows_per_img, row0);
  MH_LAUNCH_CHECK();
}

extern "C" int mh_mask_lens(const void* mask_u8, int32_t* lens, int B, int S, void* stream) {
  if (!mask_u8 || !lens || B <= 0 || S <= 0) return MH_ERR_ARG;
  hipLaunchKernelGGL(mask_lens_k, dim3(B), dim3(256), 0, as_stream(stream), (const uint8_t*)mask_u8, lens, S);
  MH_LAUNCH_CHECK();
}

extern "C" int mh_mask_unpad_index(const void* mask_u8, int64_t* fwd, int64_t* inv, int32_t* count, int B, int S, void* stream) {
  if (!mask_u8 || !fwd || !inv || !count || B <= 0 || S <= 0) return MH_ERR_ARG;
  hipLaunchKernelGGL(mask_unpad_index_k, dim3(B), dim3(256), 0, as_stream(stream), (const uint8_t*)mask_u8, fwd, inv, count, S);
  MH_LAUNCH_CHECK();
}

extern "C" int mh_check_inputs(const int64_t* ids, const int64_t* labels, const void* mask_u8, const int32_t* lens, int32_t* err,
                               int B, int S, int V, void* stream) {
  if (!err || B <= 0 || S <= 0 || V <= 0 || (mask_u8 && !lens)) return MH_ERR_ARG;
  hipLaunchKernelGGL(check_inputs_k, dim3(B), dim3(256), 0, as_stream(stream), ids, labels, (const uint8_t*)mask_u8, lens, err, S, V);
  MH_LAUNCH_CHECK();
}

extern "C" int mh_embed_splice_fwd(const int64_t* ids, const int32_t* src, const void* embed, const void* feats, void* out,
                                   int T, int d, int dt, void* stream) {
  if (!ids || !embed || !out || T <= 0 || (d & 7) || (src && !feats)) return MH_ERR_ARG;
  if (dt != MH_BF16 && dt != MH_F16) return MH_ERR_DTYPE;
  hipLaunchKernelGGL(embed_splice_fwd_k<MH_BF16>, dim3(grid_for((int64_t)T * (d >> 3))), dim3(256), 0, as_stream(stream), ids, src,
                     (const uint16_t*)embed, (const uint16_t*)feats, (uint16_t*)out, (int64_t)T, d);
  MH_LAUNCH_CHECK();
}

extern "C" int mh_embed_splice_fwd_f32(const int64_t* ids, const int32_t* src, const void* embed, const float* feats, float* out, int T, int d, int dt,
                                       void* stream) {
  if (!ids || !embed || !out || T <= 0 || (d & 7) || (src && !feats)) return MH_ERR_ARG;
  if (dt == MH_BF16)
    hipLaunchKernelGGL(embed_splice_fwd_f32_k<MH_BF16>, dim3(grid_for((int64_t)T * (d >> 3))), dim3(256), 0, as_stream(stream), ids, src,
                       (const uint16_t*)embed, feats, out, (int64_t)T, d);
  else if (dt == MH_F16)
    hipLaunchKernelGGL(embed_splice_fwd_f32_k<MH_F16>, dim3(grid_for((int64_t)T * (d >> 3))), dim3(256), 0, as_stream(stream), ids, src,
                       (const uint16_t*)embed, feats, out, (int64_t)T, d);
  else return MH_ERR_DTYPE;
  MH_LAUNCH_CHECK();
}

extern "C" int mh_embed_splice_bwd(const int64_t* ids, const int32_t* src, const void* dout, void* dfeats, float* dembed32,
                                   int T, int d, int dt, void* stream) {
  if (!ids || !dout || T <= 0 || (d & 7)) return MH_ERR_ARG;
  if (dt == MH_BF16)
    hipLaunchKernelGGL(embed_splice_bwd_k<MH_BF16>, dim3(grid_for((int64_t)T * (d >> 3))), dim3(256), 0, as_stream(stream), ids, src, (const uint16_t*)dout, (uint16_t*)dfeats, dembed32, (int64_t)T, d);
  else if (dt == MH_F16)
    hipLaunchKernelGGL(embed_splice_bwd_k<MH_F16>, dim3(grid_for((int64_t)T * (d >> 3))), dim3(256), 0, as_stream(stream), ids, src, (const uint16_t*)dout, (uint16_t*)dfeats, dembed32, (int64_t)T, d);
  else return MH_ERR_DTYPE;
  MH_LAUNCH_CHECK();
}
