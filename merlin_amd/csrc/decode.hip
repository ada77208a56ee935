// KV-cache decode step of the Llama decoder (gfx950): the S_q = 1 path behind `generate()`
// (reference: llama_mmgpt.py:114-134 prepare_inputs_for_generation; HF LlamaAttention with past_key_values,
// modeling_llama.py:243-281; eval_mmvet.py:101-120 is the caller).  Every kernel here is HBM-bound - one new token
// per sequence streams all weights (13.5 GB for Llama-7B) and the whole K/V cache once - so none of it is shaped
// into an MFMA GEMM: the rules that matter are 16-byte coalesced loads, enough waves in flight and no re-reads.
//
//   mh_gemv             y[m, n] = sum_k x[m, k] W[n, k] (+ resid[m, n]),  m <= 8 rows: ONE WAVE PER WEIGHT ROW, the
//                       row is read once as 64 lanes x 16 B per step and dotted against all m activations rows
//                       (which stay in L1/L2: m * K * 2 B <= 176 KB); fp32 accumulate, shuffle reduction.  6-16 rows: an MFMA form.
//   mh_gemv_qkv_rope    the q|k|v projection with its neighbours in the same launch: input_layernorm of the row (every block normalises
//                       it into LDS), rotate-half RoPE of q, k at the token's position, append of k, v to the cache - a wave owns the
//                       rotary pair (c, c + D/2) of one head, so lane 0 ends up with both partners.
//   mh_gemv_norm / mh_gemv_swiglu   post_attention_layernorm + gate|up projection + SwiGLU in one launch (a wave owns a gate row and
//                       its up row); fp8-weight forms of all of these (mh_gemv_fp8w*).
//   mh_decode_rope_append  (stand-alone form) rotate q, k of the new token at its own position and append k, v to the cache rows [b, pos[b]].
//   mh_attn_decode      block per (b, h, key split): pass 1 D/8 lanes per key (coalesced 256-B key rows, shuffle-reduced
//                       dot products, 4 rows in flight per thread) -> scores in LDS -> block max / sum; pass 2 lane-per-channel
//                       accumulation of p.V (coalesced value rows); split-KV partials merged by a second kernel.  Keys [0, len[b]).
// A decode step of a layer is 6 launches; each costs ~4 us of fixed time on top of its streaming, which is why the neighbours are folded in.
#include "mh_common.h"

// Weights and KV-cache rows are read ONCE per decode step by ONE CU: non-temporal loads (MI355X_MICROARCH.md "nt-weights": issued -> landed
// -18 %, a decode layer 5-10 % faster) keep them from displacing the small activation vectors every block re-reads.  A/B: -DMH_DECODE_NT=0.
#ifndef MH_DECODE_NT
#define MH_DECODE_NT 1
#endif
typedef unsigned int du32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_stream16(const void* p) {
#if MH_DECODE_NT
  const du32x4 v = __builtin_nontemporal_load((const du32x4*)p);
  return make_uint4(v[0], v[1], v[2], v[3]);
#else
  return *(const uint4*)p;
#endif
}

namespace {

// packed dot product of two 16-bit pairs with fp32 accumulate (v_dot2c_f32_bf16 / v_dot2c_f32_f16): no unpacking, half
// the VALU instructions of an fma per element - at 8 activation rows the unpack+fma form is VALU-bound, not HBM-bound
typedef __bf16 mh_bf2 __attribute__((ext_vector_type(2)));
typedef _Float16 mh_h2 __attribute__((ext_vector_type(2)));
template <int DT>
__device__ __forceinline__ float dot2_acc(uint32_t a, uint32_t b, float c) {
  if constexpr (DT == MH_BF16) return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(mh_bf2, a), __builtin_bit_cast(mh_bf2, b), c, false);
  else return __builtin_amdgcn_fdot2(__builtin_bit_cast(mh_h2, a), __builtin_bit_cast(mh_h2, b), c, false);
}

// Fused decode_rope_append of the q|k|v projection (tab == nullptr: off).  A wave then owns the ROTARY PAIR (c, c + D/2) of one head of
// one section (q, k or v) instead of two consecutive rows, so lane 0 ends up with both partners of every activation row and can rotate
// them at pos[m] (rotate-half, as rope_append_k on the stored 16-bit values) and write k / v straight into the cache rows [m, pos[m]].
struct RopeAppend {
  const float2* tab;   // [max_pos, D/2] (cos, sin)
  const int32_t* pos;  // [M] cache row the new k, v are appended at
  const int32_t* rpos; // [M] rotary position of the new token (== pos unless the prompt had padding in front of / inside it)
  uint16_t *kc, *vc;   // [M, Smax, H*D]
  int H, D, Smax;
};
template <int DT, int MM>
__device__ __forceinline__ void rope_append_store(const RopeAppend& ra, int pidx, const float (&a0)[MM], const float (&a1)[MM], uint16_t* out, int64_t ldo) {
  const int half = ra.D >> 1, per_sec = ra.H * half;
  const int sec = pidx / per_sec, rem = pidx - sec * per_sec, h = rem / half, c = rem - h * half;
  const int64_t HD = (int64_t)ra.H * ra.D, col = (int64_t)h * ra.D + c;
#pragma unroll
  for (int m = 0; m < MM; ++m) {
    float lo = ld16<DT>((uint16_t)st16<DT>(a0[m])), hi = ld16<DT>((uint16_t)st16<DT>(a1[m]));  // the projection as it would be stored
    const int p = ra.pos[m];
    if (sec < 2) {
      const float2 cs = ra.tab[(int64_t)ra.rpos[m] * half + c];
      rope_rot(lo, hi, cs.x, cs.y, lo, hi);
    }
    const uint16_t l16 = (uint16_t)st16<DT>(lo), h16 = (uint16_t)st16<DT>(hi);
    out[(int64_t)m * ldo + sec * HD + col] = l16;
    out[(int64_t)m * ldo + sec * HD + col + half] = h16;
    if (sec > 0) {
      uint16_t* dst = (sec == 1 ? ra.kc : ra.vc) + ((int64_t)m * ra.Smax + p) * HD + col;
      dst[0] = l16;
      dst[half] = h16;
    }
  }
}

// RMSNorm of MM rows (<= 8192 wide) into LDS xs[MM][K] by one 256-thread block: the chunk assignment, summation order and rounding of
// rmsnorm_fwd_row_k (norm.hip), so a projection fed from here equals rmsnorm + projection bit for bit.
template <int DT, int MM>
__device__ __forceinline__ void stage_rmsnorm(const uint16_t* __restrict__ x, int64_t ldx, const uint16_t* __restrict__ norm_w, float eps, int K,
                                              uint16_t* xs, float (*red)[4]) {
  const int tid = threadIdx.x, lane = tid & 63, nch = K >> 3;  // nch <= 1024 (launcher)
  uint4 xv[MM][4];
  float ssq[MM];
#pragma unroll
  for (int m = 0; m < MM; ++m) {
    ssq[m] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 256;
      xv[m][i] = c < nch ? *(const uint4*)(x + (int64_t)m * ldx + c * 8) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float f[8];
      unpack8<DT>(xv[m][i], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) ssq[m] += f[e] * f[e];
    }
    ssq[m] = wave_sum(ssq[m]);
    if (lane == 0) red[m][tid >> 6] = ssq[m];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + i * 256;
    if (c < nch) {
      float g[8];
      unpack8<DT>(*(const uint4*)(norm_w + c * 8), g);
#pragma unroll
      for (int m = 0; m < MM; ++m) {
        const float r = rsqrtf(((red[m][0] + red[m][1]) + (red[m][2] + red[m][3])) / (float)K + eps);
        float f[8];
        unpack8<DT>(xv[m][i], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = f[e] * r * g[e];
        *(uint4*)(xs + m * K + c * 8) = pack8<DT>(f);
      }
    }
  }
  __syncthreads();
}

// ROWS weight rows per wave, 4 waves per block.  LDSX: the block first stages a K-chunk of the MM activation rows in
// LDS (MM x 4096 x 2 B = 64 KiB at MM = 8) and every wave reads it from there: without it each wave re-fetches all
// activations through L1/L2 (8x the weight bytes at MM = 8; measured 1.9 TB/s of weights instead of 5).
constexpr int GEMV_KC = 2048;  // 32 KiB at MM = 8: four blocks (16 waves) per CU keep enough weight loads in flight
// NSTEP 512-element steps of every weight row are requested before any is consumed (2; 8 measured the same at N = 4096, where a
// launch is 2048 short-lived waves: profiles/r02_decode_bench.txt).
// NORM: x is the input of the RMSNorm that precedes the projection (HF LlamaRMSNorm, weight norm_w): every block normalises the MM
// rows itself (8 KB each, L2-resident) into LDS - same chunk assignment, summation order and rounding as rmsnorm_fwd_row_k, so the
// result equals the two launches bit for bit - instead of a separate kernel per norm (65 per decode step).
template <int DT, int MM, int ROWS, bool LDSX, int NSTEP, bool NORM = false>
__global__ __launch_bounds__(256) void gemv_k(const uint16_t* __restrict__ x, int64_t ldx, const uint16_t* __restrict__ W,
                                              int64_t ldw, void* __restrict__ out, int64_t ldo, const uint16_t* __restrict__ resid,
                                              int64_t ldr, int N, int K, int out_f32, int swi_ff, const uint16_t* __restrict__ norm_w,
                                              float eps, RopeAppend ra) {
  extern __shared__ __attribute__((aligned(16))) uint16_t xs[];  // [MM][GEMV_KC] when LDSX, [MM][K] when NORM
  __shared__ float red[NORM ? MM : 1][4];
  const int xstride = NORM ? K : GEMV_KC;
  const int lane = threadIdx.x & 63;
  // swi_ff > 0 (fused SwiGLU of the gate|up projection, W = [2 ff, K], N = ff outputs): the wave's rows are ROWS/2 gate rows n and
  // the matching up rows ff + n, and it writes act[n] = silu(gate) * up of the ROUNDED 16-bit gate / up values (= mh_swiglu_fwd on
  // the stored projection)
  const int NR = swi_ff > 0 ? ROWS / 2 : ROWS;  // output columns per wave
  const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * NR;
  float acc[ROWS][MM];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int m = 0; m < MM; ++m) acc[r][m] = 0.f;
  const uint16_t* wrow[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    int wr_ = swi_ff > 0 ? (r < NR ? min(n0 + r, N - 1) : swi_ff + min(n0 + r - NR, N - 1)) : min(n0 + r, N - 1);
    if (ROWS == 2 && ra.tab) {  // rotary pair (c, c + D/2) of pair index n0 / 2 (N = 3 H D outputs, N / 2 pairs)
      const int half = ra.D >> 1, pidx = min(n0 >> 1, (N >> 1) - 1), per_sec = ra.H * half;
      const int sec = pidx / per_sec, rem = pidx - sec * per_sec, h = rem / half;
      wr_ = sec * ra.H * ra.D + h * ra.D + (rem - h * half) + r * half;
    }
    wrow[r] = W + (int64_t)wr_ * ldw;
  }
  for (int kc = 0; kc < K; kc += GEMV_KC) {
    const int klen = (LDSX && !NORM) ? min(GEMV_KC, K - kc) : K;  // (without LDS staging, or with the whole row staged, the K loop is not chunked)
    if constexpr (NORM) {
      stage_rmsnorm<DT, MM>(x, ldx, norm_w, eps, K, xs, red);
    } else if constexpr (LDSX) {
      if (kc) __syncthreads();
      for (int i = threadIdx.x * 8; i < MM * klen; i += 256 * 8) {
        const int m = i / klen, k = i - m * klen;
        *(uint4*)(xs + m * GEMV_KC + k) = *(const uint4*)(x + (int64_t)m * ldx + kc + k);
      }
      __syncthreads();
    }
    if (n0 < N) {
      for (int k0 = lane * 8; k0 < klen; k0 += 512 * NSTEP) {
        uint4 wv[NSTEP][ROWS];
#pragma unroll
        for (int h2 = 0; h2 < NSTEP; ++h2)
#pragma unroll
          for (int r = 0; r < ROWS; ++r)
            wv[h2][r] = (k0 + h2 * 512 < klen) ? ld_stream16(wrow[r] + kc + k0 + h2 * 512) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int h2 = 0; h2 < NSTEP; ++h2) {
          const int kk = k0 + h2 * 512;
          if (kk >= klen) break;
#pragma unroll
          for (int m = 0; m < MM; ++m) {
            uint4 xv;
            if constexpr (LDSX || NORM) xv = *(const uint4*)(xs + m * xstride + kk);
            else xv = *(const uint4*)(x + (int64_t)m * ldx + kc + kk);
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
              float a = acc[r][m];
              a = dot2_acc<DT>(wv[h2][r].x, xv.x, a);
              a = dot2_acc<DT>(wv[h2][r].y, xv.y, a);
              a = dot2_acc<DT>(wv[h2][r].z, xv.z, a);
              a = dot2_acc<DT>(wv[h2][r].w, xv.w, a);
              acc[r][m] = a;
            }
          }
        }
      }
    }
    if constexpr (!LDSX || NORM) break;
  }
  if (n0 >= N) return;
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int m = 0; m < MM; ++m) acc[r][m] = wave_sum(acc[r][m]);
  if constexpr (ROWS == 2) {
    if (ra.tab) {
      if (lane == 0) rope_append_store<DT, MM>(ra, n0 >> 1, acc[0], acc[1], (uint16_t*)out, ldo);
      return;
    }
  }
  if (lane == 0 && swi_ff > 0) {
#pragma unroll
    for (int r = 0; r < ROWS / 2; ++r) {
      const int n = n0 + r;
      if (n >= N) break;
#pragma unroll
      for (int m = 0; m < MM; ++m) {
        const float g_ = ld16<DT>((uint16_t)st16<DT>(acc[r][m])), u_ = ld16<DT>((uint16_t)st16<DT>(acc[r + ROWS / 2][m]));
        ((uint16_t*)out)[(int64_t)m * ldo + n] = (uint16_t)st16<DT>(swiglu_fwd1(g_, u_));
      }
    }
    return;
  }
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int n = n0 + r;
      if (n >= N) break;
#pragma unroll
      for (int m = 0; m < MM; ++m) {
        float v = acc[r][m];
        if (resid) v += ld16<DT>(resid[(int64_t)m * ldr + n]);
        if (out_f32) ((float*)out)[(int64_t)m * ldo + n] = v;
        else ((uint16_t*)out)[(int64_t)m * ldo + n] = (uint16_t)st16<DT>(v);
      }
    }
  }
}

// 1-2 activation rows and a SMALL N (the o / down projections: 4096 rows): one wave per ROWS weight rows gives 2048 waves for 256 CUs,
// each streaming its rows for the whole of K - the launch is a ramp-up and a tail.  Here the four waves of a block share the same
// ROWS rows and split K in four, i.e. four times as many waves with a quarter of the work each (all 32 wave slots of a CU busy); the
// four partial sums meet in LDS and are added in a fixed order.
template <int DT, int MM, int ROWS>
__global__ __launch_bounds__(256) void gemv_ks_k(const uint16_t* __restrict__ x, int64_t ldx, const uint16_t* __restrict__ W, int64_t ldw,
                                                 void* __restrict__ out, int64_t ldo, const uint16_t* __restrict__ resid, int64_t ldr, int N,
                                                 int K, int out_f32) {
  __shared__ float red[4][ROWS][MM];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * ROWS;
  const int kq = (((K >> 3) + 3) >> 2) << 3;  // elements per wave (a multiple of 8)
  const int kb = wave * kq, ke = min(K, kb + kq);
  float acc[ROWS][MM];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int m = 0; m < MM; ++m) acc[r][m] = 0.f;
  const uint16_t* wrow[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) wrow[r] = W + (int64_t)min(n0 + r, N - 1) * ldw;
  for (int k0 = kb + lane * 8; k0 < ke; k0 += 1024) {
    uint4 wv[2][ROWS];
    const bool two = k0 + 512 < ke;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) wv[0][r] = ld_stream16(wrow[r] + k0);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) wv[1][r] = two ? ld_stream16(wrow[r] + k0 + 512) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      if (h2 == 1 && !two) break;
#pragma unroll
      for (int m = 0; m < MM; ++m) {
        const uint4 xv = *(const uint4*)(x + (int64_t)m * ldx + k0 + h2 * 512);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          float a = acc[r][m];
          a = dot2_acc<DT>(wv[h2][r].x, xv.x, a);
          a = dot2_acc<DT>(wv[h2][r].y, xv.y, a);
          a = dot2_acc<DT>(wv[h2][r].z, xv.z, a);
          a = dot2_acc<DT>(wv[h2][r].w, xv.w, a);
          acc[r][m] = a;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int m = 0; m < MM; ++m) {
      const float t = wave_sum(acc[r][m]);
      if (lane == 0) red[wave][r][m] = t;
    }
  __syncthreads();
  if (threadIdx.x < ROWS * MM) {
    const int r = threadIdx.x / MM, m = threadIdx.x % MM, n = n0 + r;
    if (n < N) {
      float v = (red[0][r][m] + red[1][r][m]) + (red[2][r][m] + red[3][r][m]);
      if (resid) v += ld16<DT>(resid[(int64_t)m * ldr + n]);
      if (out_f32) ((float*)out)[(int64_t)m * ldo + n] = v;
      else ((uint16_t*)out)[(int64_t)m * ldo + n] = (uint16_t)st16<DT>(v);
    }
  }
}

// ---- fp8 (OCP e4m3) weights with one fp32 scale per 128 consecutive k (BASELINE cfg 5's weight format), bf16/f16
// activations: the decode step is weight-bandwidth-bound, so halving the weight bytes is worth ~2x on the GEMVs.
// quant: q[n, k] = fp8(w[n, k] / s[n, k/128]),  s = max|w| over the block / 448 (1 if the block is all zero).
template <int DT>
__global__ __launch_bounds__(256) void quant_fp8_b128_k(const uint16_t* __restrict__ w, int64_t ldw, uint8_t* __restrict__ q,
                                                        float* __restrict__ sc, int N, int K) {
  const int nb = (K + 127) / 128;
  const int64_t blk = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);  // 16 lanes per 128-element block
  if (blk >= (int64_t)N * nb) return;
  const int n = (int)(blk / nb), kb = (int)(blk % nb);
  const int k0 = kb * 128 + (threadIdx.x & 15) * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  if (k0 < K) unpack8<DT>(*(const uint4*)(w + (int64_t)n * ldw + k0), v);
  float mx = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(v[e]));
#pragma unroll
  for (int o2 = 8; o2 > 0; o2 >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o2, 64));
  const float s = mx > 0.f ? mx * (1.0f / 448.0f) : 1.0f;
  const float inv = 1.0f / s;
  if ((threadIdx.x & 15) == 0) sc[(int64_t)n * nb + kb] = s;
  if (k0 < K) {
    int p0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, 0, false);
    p0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, p0, true);
    int p1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, 0, false);
    p1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, p1, true);
    *(uint2*)(q + (int64_t)n * K + k0) = make_uint2((unsigned)p0, (unsigned)p1);
  }
}

// two e4m3 values (the low or the high half of a dword) -> one packed 16-bit pair, in ONE instruction (gfx950 v_cvt_scalef32_pk_*_fp8 with
// scale 1: e4m3 values are exact in bf16 and fp16); a dword of 4 fp8 -> 2 packed words
template <int DT>
__device__ __forceinline__ void fp8x4_to_pk16(uint32_t p, uint32_t& lo, uint32_t& hi) {
  if constexpr (DT == MH_BF16) {
    lo = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(p, 1.0f, false));
    hi = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(p, 1.0f, true));
  } else {
    lo = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(p, 1.0f, false));
    hi = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(p, 1.0f, true));
  }
}
__device__ __forceinline__ void fp8x4_to_f32(uint32_t p, float* f) {
  typedef float f2_ __attribute__((ext_vector_type(2)));
  const f2_ lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)p, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)p, true);
  f[0] = lo[0]; f[1] = lo[1]; f[2] = hi[0]; f[3] = hi[1];
}

// y[m, n] = sum_kb s[n, kb] * sum_{k in block} q[n, k] x[m, k] (+ resid): one wave per weight row, 16 fp8 (16 B) per lane
// and step (a lane's 16 values lie inside one 128-block), fp32 accumulate.  K % 16 == 0.
// (NORM / swi_ff: the fused RMSNorm and SwiGLU of gemv_k, same semantics)
template <int DT, int MM, int ROWS, bool LDSX, bool NORM = false>
__global__ __launch_bounds__(256) void gemv_fp8w_k(const uint16_t* __restrict__ x, int64_t ldx, const uint8_t* __restrict__ q,
                                                   const float* __restrict__ sc, void* __restrict__ out, int64_t ldo,
                                                   const uint16_t* __restrict__ resid, int64_t ldr, int N, int K, int out_f32, int swi_ff,
                                                   const uint16_t* __restrict__ norm_w, float eps, RopeAppend ra) {
  extern __shared__ __attribute__((aligned(16))) uint16_t xs8[];  // [MM][GEMV_KC] when LDSX (as in gemv_k), [MM][K] when NORM
  __shared__ float red[NORM ? MM : 1][4];
  const int xstride = NORM ? K : GEMV_KC;
  const int lane = threadIdx.x & 63;
  const int NR = swi_ff > 0 ? ROWS / 2 : ROWS;  // output columns per wave
  const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * NR;
  const int nb = (K + 127) / 128;
  float acc[ROWS][MM];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int m = 0; m < MM; ++m) acc[r][m] = 0.f;
  const uint8_t* qrow[ROWS];
  const float* srow[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    int n = swi_ff > 0 ? (r < NR ? min(n0 + r, N - 1) : swi_ff + min(n0 + r - NR, N - 1)) : min(n0 + r, N - 1);
    if (ROWS == 2 && ra.tab) {  // rotary pair (as in gemv_k)
      const int half = ra.D >> 1, pidx = min(n0 >> 1, (N >> 1) - 1), per_sec = ra.H * half;
      const int sec = pidx / per_sec, rem = pidx - sec * per_sec, h = rem / half;
      n = sec * ra.H * ra.D + h * ra.D + (rem - h * half) + r * half;
    }
    qrow[r] = q + (int64_t)n * K;
    srow[r] = sc + (int64_t)n * nb;
  }
  for (int kc = 0; kc < K; kc += GEMV_KC) {
    const int klen = NORM ? K : min(GEMV_KC, K - kc);
    if constexpr (NORM) {
      stage_rmsnorm<DT, MM>(x, ldx, norm_w, eps, K, xs8, red);
    } else if constexpr (LDSX) {
      if (kc) __syncthreads();
      for (int i = threadIdx.x * 8; i < MM * klen; i += 256 * 8) {
        const int m = i / klen, k = i - m * klen;
        *(uint4*)(xs8 + m * GEMV_KC + k) = *(const uint4*)(x + (int64_t)m * ldx + kc + k);
      }
      __syncthreads();
    }
    if (n0 < N) {
      for (int k0 = lane * 16; k0 < klen; k0 += 1024) {
        uint4 qv[ROWS];
        float s[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          qv[r] = ld_stream16(qrow[r] + kc + k0);
          s[r] = srow[r][(kc + k0) >> 7];
        }
        float p[ROWS][MM];
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
          for (int m = 0; m < MM; ++m) p[r][m] = 0.f;
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {  // 8 values at a time: the weights become packed 16-bit pairs (exact) and meet the activations in
                                             // dot2 instructions, as in gemv_k (4 cvt + 4 x MM dot2 per row instead of 4 cvt + 8 x MM fma)
          uint32_t w[ROWS][4];
#pragma unroll
          for (int r = 0; r < ROWS; ++r) {
            fp8x4_to_pk16<DT>(hlf ? qv[r].z : qv[r].x, w[r][0], w[r][1]);
            fp8x4_to_pk16<DT>(hlf ? qv[r].w : qv[r].y, w[r][2], w[r][3]);
          }
#pragma unroll
          for (int m = 0; m < MM; ++m) {
            uint4 xa;
            if constexpr (LDSX || NORM) xa = *(const uint4*)(xs8 + m * xstride + k0 + 8 * hlf);
            else xa = *(const uint4*)(x + (int64_t)m * ldx + kc + k0 + 8 * hlf);
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
              float a = p[r][m];
              a = dot2_acc<DT>(w[r][0], xa.x, a);
              a = dot2_acc<DT>(w[r][1], xa.y, a);
              a = dot2_acc<DT>(w[r][2], xa.z, a);
              a = dot2_acc<DT>(w[r][3], xa.w, a);
              p[r][m] = a;
            }
          }
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
          for (int m = 0; m < MM; ++m) acc[r][m] = fmaf(s[r], p[r][m], acc[r][m]);
      }
    }
    if constexpr (NORM) break;
  }
  if (n0 >= N) return;
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int m = 0; m < MM; ++m) acc[r][m] = wave_sum(acc[r][m]);
  if constexpr (ROWS == 2) {
    if (ra.tab) {
      if (lane == 0) rope_append_store<DT, MM>(ra, n0 >> 1, acc[0], acc[1], (uint16_t*)out, ldo);
      return;
    }
  }
  if (lane == 0 && swi_ff > 0) {
#pragma unroll
    for (int r = 0; r < ROWS / 2; ++r) {
      const int n = n0 + r;
      if (n >= N) break;
#pragma unroll
      for (int m = 0; m < MM; ++m) {
        const float g_ = ld16<DT>((uint16_t)st16<DT>(acc[r][m])), u_ = ld16<DT>((uint16_t)st16<DT>(acc[r + ROWS / 2][m]));
        ((uint16_t*)out)[(int64_t)m * ldo + n] = (uint16_t)st16<DT>(swiglu_fwd1(g_, u_));
      }
    }
    return;
  }
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int n = n0 + r;
      if (n >= N) break;
#pragma unroll
      for (int m = 0; m < MM; ++m) {
        float v = acc[r][m];
        if (resid) v += ld16<DT>(resid[(int64_t)m * ldr + n]);
        if (out_f32) ((float*)out)[(int64_t)m * ldo + n] = v;
        else ((uint16_t*)out)[(int64_t)m * ldo + n] = (uint16_t)st16<DT>(v);
      }
    }
  }
}


// ---- 3..16 activation rows: the matrix cores do the multiplies --------------------------------------------------------------------
// gemv_ks_k with fp8 weights: the four waves of a block share ROWS rows and split K (a multiple of 128 per wave, so a lane's 16 values
// stay inside one scale block); partial sums meet in LDS, fixed order.
template <int DT, int MM, int ROWS>
__global__ __launch_bounds__(256) void gemv_fp8w_ks_k(const uint16_t* __restrict__ x, int64_t ldx, const uint8_t* __restrict__ q,
                                                      const float* __restrict__ sc, void* __restrict__ out, int64_t ldo,
                                                      const uint16_t* __restrict__ resid, int64_t ldr, int N, int K, int out_f32) {
  __shared__ float red[4][ROWS][MM];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * ROWS;
  const int nb = (K + 127) / 128;
  const int kq = ((nb + 3) >> 2) << 7;  // elements per wave: whole 128-blocks
  const int kb = wave * kq, ke = min(K, kb + kq);
  float acc[ROWS][MM];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int m = 0; m < MM; ++m) acc[r][m] = 0.f;
  const uint8_t* qrow[ROWS];
  const float* srow[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int n = min(n0 + r, N - 1);
    qrow[r] = q + (int64_t)n * K;
    srow[r] = sc + (int64_t)n * nb;
  }
  for (int k0 = kb + lane * 16; k0 < ke; k0 += 1024) {
    uint4 qv[ROWS];
    float s[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      qv[r] = ld_stream16(qrow[r] + k0);
      s[r] = srow[r][k0 >> 7];
    }
    float p[ROWS][MM];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int m = 0; m < MM; ++m) p[r][m] = 0.f;
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
      uint32_t w[ROWS][4];
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        fp8x4_to_pk16<DT>(hlf ? qv[r].z : qv[r].x, w[r][0], w[r][1]);
        fp8x4_to_pk16<DT>(hlf ? qv[r].w : qv[r].y, w[r][2], w[r][3]);
      }
#pragma unroll
      for (int m = 0; m < MM; ++m) {
        const uint4 xa = *(const uint4*)(x + (int64_t)m * ldx + k0 + 8 * hlf);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          float a = p[r][m];
          a = dot2_acc<DT>(w[r][0], xa.x, a);
          a = dot2_acc<DT>(w[r][1], xa.y, a);
          a = dot2_acc<DT>(w[r][2], xa.z, a);
          a = dot2_acc<DT>(w[r][3], xa.w, a);
          p[r][m] = a;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int m = 0; m < MM; ++m) acc[r][m] = fmaf(s[r], p[r][m], acc[r][m]);
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int m = 0; m < MM; ++m) {
      const float t = wave_sum(acc[r][m]);
      if (lane == 0) red[wave][r][m] = t;
    }
  __syncthreads();
  if (threadIdx.x < ROWS * MM) {
    const int r = threadIdx.x / MM, m = threadIdx.x % MM, n = n0 + r;
    if (n < N) {
      float v = (red[0][r][m] + red[1][r][m]) + (red[2][r][m] + red[3][r][m]);
      if (resid) v += ld16<DT>(resid[(int64_t)m * ldr + n]);
      if (out_f32) ((float*)out)[(int64_t)m * ldo + n] = v;
      else ((uint16_t*)out)[(int64_t)m * ldo + n] = (uint16_t)st16<DT>(v);
    }
  }
}

// At batch 1-2 the one-wave-per-row kernel above is HBM-bound; from ~4 rows on its VALU work (rows x 8 dot2 per 16 B of weights, x 2
// more with fp8 dequantisation) is what limits it.  Here a block owns 16 x RG weight rows and its NW waves split K: every lane loads
// 16 B of one weight row STRAIGHT INTO the MFMA's B-operand registers (lane l: row l & 15, k-chunk l >> 4), and the <= 16 activation
// rows are the A operand, ALSO loaded straight from global memory (L2-resident: M x K x 2 B): every k step is consumed by exactly one
// wave of the block, so an LDS copy of the activations shared nothing and cost two barriers and a dependent round trip per 2 K of k
// (the first form of this kernel: 2.5 TB/s at N = 4096).  One 16x16x32 MFMA per 1 KB of weights replaces 16 x M dot2 instructions:
// the kernel is HBM-bound for any M <= 16.  fp8 weights (16 values per 16 B, per-128-block fp32 scales) are converted to 16-bit in
// registers (16 cvt per load instead of 16 x M fma) and their block's partial product is scaled once per load step.  The K loop runs in
// passes of NW waves x NS steps whose loads are all requested before the first MFMA.  Partial tiles of the NW waves are summed through
// LDS in a fixed order.
// RG = 16-row groups per block (1, 2 or 4): a block re-reads the whole activation matrix (M x K) from L2 whatever it does with it, so
// at 16 rows per block that traffic equals the weight stream at M = 16 and the kernel is L2-bound (measured 2.9 TB/s); 64 rows per
// block share one A fragment per step between four B fragments.  Large N only: N / 64 blocks must still fill the chip.
// NW = 8, or 16 for small N (N = 4096: 256 blocks x 16 waves, i.e. twice the loads in flight at the start).
// PAIR (RG = 2): the block's second row group is the PARTNER of the first instead of the next 16 rows, and wave 0's epilogue combines
// them on the rounded 16-bit values, exactly as the separate launches would: 1 = SwiGLU of the gate|up projection (group 1 = up rows
// ff + n; N = ff outputs act = silu(gate) * up), 2 = RoPE + K/V append of the q|k|v projection (group 1 = channel c + D/2 of the same
// head; a block is 16 channels c of one head of one section; D/2 a multiple of 16).
template <int DT, bool FP8W, int RG, int NW, int PAIR = 0>
__global__ __launch_bounds__(NW * 64) void gemv_mfma_k(const uint16_t* __restrict__ x, int64_t ldx, const void* __restrict__ Wv, int64_t ldw,
                                                       const float* __restrict__ wsc, void* __restrict__ out, int64_t ldo,
                                                       const uint16_t* __restrict__ resid, int64_t ldr, int M, int N, int K, int out_f32,
                                                       int swi_ff, RopeAppend ra) {
  static_assert(PAIR == 0 || RG == 2, "paired epilogues take two row groups");
  __shared__ __attribute__((aligned(16))) float red[NW * RG * 64 * 4];
  constexpr int KS = FP8W ? 64 : 32;                      // k per step (16 B of one weight row per lane, 4 lanes per row)
  constexpr int NS = FP8W ? (RG == 1 ? 6 : 4) : (RG == 4 ? 4 : 8);  // steps per wave per pass (registers)
  constexpr int KP = NW * NS * KS;                        // k per pass
  constexpr int ES = FP8W ? 1 : 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  int n0 = blockIdx.x * 16 * RG, goff = 16, nlim = N;  // first row, distance between the row groups, row count of W
  if constexpr (PAIR == 1) { n0 = blockIdx.x * 16; goff = swi_ff; nlim = N + swi_ff; }
  if constexpr (PAIR == 2) {
    const int half = ra.D >> 1, spb = half >> 4, slot = blockIdx.x / spb;  // slot = section * H + head
    n0 = slot * ra.D + (blockIdx.x - slot * spb) * 16;
    goff = half;
  }
  const bool arow = j < M;
  f32x4_t acc[RG];
  const uint8_t* wr[RG];   // byte pointers: element size 1 (fp8) or 2
  const float* sr[RG];
  const int nb = (K + 127) / 128;
#pragma unroll
  for (int g = 0; g < RG; ++g) {
    const int n = PAIR == 1 ? min(n0 + j, N - 1) + g * goff : min(n0 + goff * g + j, nlim - 1);
    acc[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    wr[g] = (const uint8_t*)Wv + ((int64_t)n * ldw + kq * (KS / 4)) * ES;
    sr[g] = FP8W ? wsc + (int64_t)n * nb : nullptr;
  }
  const uint16_t* xr = x + (int64_t)min(j, M - 1) * ldx + kq * (KS / 4);
  for (int kp = 0; kp < K; kp += KP) {
    uint4 wv[RG][NS], xv[NS][FP8W ? 2 : 1];
    float sv[FP8W ? RG : 1][FP8W ? NS : 1];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int k = kp + (wave + NW * u) * KS;
      const bool ok = k < K;
#pragma unroll
      for (int g = 0; g < RG; ++g) {
        wv[g][u] = ok ? ld_stream16(wr[g] + (int64_t)k * ES) : make_uint4(0, 0, 0, 0);
        if constexpr (FP8W) sv[g][u] = ok ? sr[g][k >> 7] : 0.f;
      }
      xv[u][0] = (ok && arow) ? *(const uint4*)(xr + k) : make_uint4(0, 0, 0, 0);
      if constexpr (FP8W) xv[u][1] = (ok && arow) ? *(const uint4*)(xr + k + 8) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      if (kp + (wave + NW * u) * KS >= K) break;
      if constexpr (!FP8W) {
#pragma unroll
        for (int g = 0; g < RG; ++g) acc[g] = mfma16<DT>(xv[u][0], wv[g][u], acc[g]);
      } else {
#pragma unroll
        for (int g = 0; g < RG; ++g) {
          uint4 b0, b1;  // 16 e4m3 -> 2 x 8 packed 16-bit values (exact), one convert per pair
          fp8x4_to_pk16<DT>(wv[g][u].x, b0.x, b0.y); fp8x4_to_pk16<DT>(wv[g][u].y, b0.z, b0.w);
          fp8x4_to_pk16<DT>(wv[g][u].z, b1.x, b1.y); fp8x4_to_pk16<DT>(wv[g][u].w, b1.z, b1.w);
          f32x4_t part = {0.f, 0.f, 0.f, 0.f};
          part = mfma16<DT>(xv[u][0], b0, part);
          part = mfma16<DT>(xv[u][1], b1, part);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[g][q] = fmaf(sv[g][u], part[q], acc[g][q]);
        }
      }
    }
  }
  // D lane l reg r = D[m = 4 * (l >> 4) + r][n = l & 15]: sum the NW waves' tiles in wave order
#pragma unroll
  for (int g = 0; g < RG; ++g) *(f32x4_t*)(red + ((wave * RG + g) * 64 + lane) * 4) = acc[g];
  __syncthreads();
  if constexpr (PAIR != 0) {
    if (wave == 0) {
      f32x4_t s0 = *(const f32x4_t*)(red + lane * 4), s1 = *(const f32x4_t*)(red + (64 + lane) * 4);
#pragma unroll 4
      for (int w = 1; w < NW; ++w) {
        const f32x4_t t0 = *(const f32x4_t*)(red + ((w * 2) * 64 + lane) * 4), t1 = *(const f32x4_t*)(red + ((w * 2 + 1) * 64 + lane) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { s0[r] += t0[r]; s1[r] += t1[r]; }
      }
      uint16_t* o16 = (uint16_t*)out;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 4 * kq + r;
        if (m >= M) continue;
        float lo = ld16<DT>((uint16_t)st16<DT>(s0[r])), hi = ld16<DT>((uint16_t)st16<DT>(s1[r]));  // the projection as it would be stored
        if constexpr (PAIR == 1) {
          if (n0 + j < N) o16[(int64_t)m * ldo + n0 + j] = (uint16_t)st16<DT>(swiglu_fwd1(lo, hi));
        } else {
          const int half = ra.D >> 1, slot = n0 / ra.D, sec = slot / ra.H, c = n0 - slot * ra.D + j, p = ra.pos[m];
          const int64_t HD = (int64_t)ra.H * ra.D, col = (int64_t)(slot - sec * ra.H) * ra.D + c;
          if (sec < 2) {
            const float2 cs = ra.tab[(int64_t)ra.rpos[m] * half + c];
            rope_rot(lo, hi, cs.x, cs.y, lo, hi);
          }
          const uint16_t l16 = (uint16_t)st16<DT>(lo), h16 = (uint16_t)st16<DT>(hi);
          o16[(int64_t)m * ldo + sec * HD + col] = l16;
          o16[(int64_t)m * ldo + sec * HD + col + half] = h16;
          if (sec > 0) {
            uint16_t* dst = (sec == 1 ? ra.kc : ra.vc) + ((int64_t)m * ra.Smax + p) * HD + col;
            dst[0] = l16;
            dst[half] = h16;
          }
        }
      }
    }
    return;
  }
  if (wave < RG) {
    const int g = wave;
    f32x4_t s4 = *(const f32x4_t*)(red + (g * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const f32x4_t t = *(const f32x4_t*)(red + ((w * RG + g) * 64 + lane) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) s4[r] += t[r];
    }
    const int n = n0 + 16 * g + j;
    if (n < N) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 4 * kq + r;
        if (m >= M) continue;
        float v = s4[r];
        if (resid) v += ld16<DT>(resid[(int64_t)m * ldr + n]);
        if (out_f32) ((float*)out)[(int64_t)m * ldo + n] = v;
        else ((uint16_t*)out)[(int64_t)m * ldo + n] = (uint16_t)st16<DT>(v);
      }
    }
  }
}

static int g_gemv_mfma_nw16 = 1;  // 16 waves per block at small N (A-B switch: mh_gemv_mfma_wide)
extern "C" void mh_gemv_mfma_wide(int on) { g_gemv_mfma_nw16 = on ? 1 : 0; }

template <int DT, bool FP8W>
static int launch_gemv_mfma(const void* x, int64_t ldx, const void* W, int64_t ldw, const float* wsc, void* out, int64_t ldo, const void* resid,
                            int64_t ldr, int M, int N, int K, int out_f32, int swi_ff, const RopeAppend& ra, hipStream_t st) {
  // rows per block: as many as still give ~1.5+ blocks per CU: measured, 64-row blocks win at N = 32 064 (501 blocks) and lose at
  // N = 22 016 (344 blocks: an uneven second block per CU)
#define MH_GM(RG_, NW_, PAIR_, BLOCKS_)                                                                                                \
  hipLaunchKernelGGL((gemv_mfma_k<DT, FP8W, RG_, NW_, PAIR_>), dim3(BLOCKS_), dim3(64 * NW_), 0, st, (const uint16_t*)x, ldx, W, ldw, wsc, \
                     out, ldo, (const uint16_t*)resid, ldr, M, N, K, out_f32, swi_ff, ra)
  if (swi_ff) MH_GM(2, 8, 1, (N + 15) / 16);           // N = ff outputs
  else if (ra.tab) {                                   // N = 3 H D rows, 16 rotary pairs per block (N = 12 288: 384 blocks)
    if (g_gemv_mfma_nw16) MH_GM(2, 16, 2, N / 32); else MH_GM(2, 8, 2, N / 32);
  }
  else if (N >= 30000) MH_GM(4, 8, 0, (N + 63) / 64);
  else if (N >= 12000) MH_GM(2, 8, 0, (N + 31) / 32);
  else if (N <= 8192 && g_gemv_mfma_nw16) MH_GM(1, 16, 0, (N + 15) / 16);
  else MH_GM(1, 8, 0, (N + 15) / 16);
#undef MH_GM
  MH_LAUNCH_CHECK();
}

// qkv [B, 3, H, D] of the new tokens; tab [max_pos, D/2] (cos, sin); kc, vc [B, Smax, H*D]
template <int DT>
__global__ __launch_bounds__(256) void rope_append_k(uint16_t* __restrict__ qkv, const float2* __restrict__ tab,
                                                     const int32_t* __restrict__ pos, const int32_t* __restrict__ rpos,
                                                     uint16_t* __restrict__ kc, uint16_t* __restrict__ vc, int B, int H, int D, int Smax) {
  const int half = D >> 1, vph = half >> 3;
  const int64_t total = (int64_t)B * H * vph;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int v = (int)(i % vph);
  const int h = (int)((i / vph) % H);
  const int b = (int)(i / ((int64_t)vph * H));
  const int p = pos[b];
  const float2* tb = tab + (int64_t)rpos[b] * half + v * 8;
  const int64_t hd = (int64_t)h * D + v * 8;
  uint16_t* qb = qkv + (int64_t)b * 3 * H * D + hd;
  uint16_t* kb = qb + (int64_t)H * D;
  const uint16_t* vb = kb + (int64_t)H * D;
  uint16_t* kdst = kc + ((int64_t)b * Smax + p) * H * D + hd;
  uint16_t* vdst = vc + ((int64_t)b * Smax + p) * H * D + hd;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    uint16_t* base = which ? kb : qb;
    float lo[8], hi[8];
    unpack8<DT>(*(const uint4*)base, lo);
    unpack8<DT>(*(const uint4*)(base + half), hi);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      rope_rot(lo[k], hi[k], tb[k].x, tb[k].y, lo[k], hi[k]);
    }
    const uint4 plo = pack8<DT>(lo), phi = pack8<DT>(hi);
    *(uint4*)base = plo;
    *(uint4*)(base + half) = phi;
    if (which) {
      *(uint4*)kdst = plo;
      *(uint4*)(kdst + half) = phi;
    }
  }
  *(uint4*)vdst = *(const uint4*)vb;
  *(uint4*)(vdst + half) = *(const uint4*)(vb + half);
}

__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// ticket counters of the split-KV merge, one per (b, h): zero at load, left zero by every launch (one decode stream at a time)
__device__ unsigned g_decode_tickets[16384];

// q [B, ldq] (head h at h*D); kc, vc [B, Smax, H*D]; out [B, H*D]; keys [0, len[b]).  D in {64, 128}.
// Split-KV ("flash decoding"): with B*H blocks only (32 at batch 1) the cache streams at a few % of HBM speed, so
// `splits` blocks share one (b, h), each takes `chunk` keys and leaves (unnormalised o[D], max, sum) in `ws`;
// attn_decode_combine_k merges them as a second launch (cnt == NULL, the default) - or the last of the blocks to finish does
// (cnt != NULL: mh_attn_decode_fused_merge(1), an A/B arm that measured no faster).  splits == 1 writes the normalised result directly.
template <int DT, int D>
__global__ __launch_bounds__(256) void attn_decode_k(const uint16_t* __restrict__ q, int64_t ldq, const uint16_t* __restrict__ kc,
                                                     const uint16_t* __restrict__ vc, uint16_t* __restrict__ out,
                                                     const int32_t* __restrict__ lens, int H, int Smax, float scale_log2,
                                                     int splits, int chunk, float* __restrict__ ws, unsigned* __restrict__ cnt) {
  extern __shared__ float sc[];  // [chunk] scores, then [G][D] partial outputs
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sp = blockIdx.x % splits;
  const int h = (blockIdx.x / splits) % H, b = blockIdx.x / (splits * H);
  const int key0 = sp * chunk;
  const int len = max(0, min(min(lens[b], Smax) - key0, chunk));  // keys [key0, key0 + len) of this split
  kc += (int64_t)key0 * H * D;
  vc += (int64_t)key0 * H * D;
  const int64_t HD = (int64_t)H * D;
  // pass 1: D/8 lanes per key (one 16-byte piece each: a key row is one coalesced 256-byte read), 256*8/D keys per
  // block iteration; the partial dot products are summed across the lane group by shuffles
  constexpr int OCT1 = D / 8, KPI = 256 / OCT1;
  const int kpart = tid % OCT1, ksub = tid / OCT1;
  float q8[8];
  unpack8<DT>(*(const uint4*)(q + (int64_t)b * ldq + (int64_t)h * D + kpart * 8), q8);
  float mx = -1e30f;
  // (UNR key rows requested per thread before any is used: the cache is streamed, so what limits the rate is bytes in flight)
  constexpr int UNR = 4;
  for (int j0 = 0; j0 < len; j0 += KPI * UNR) {
    uint4 kraw[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = j0 + u * KPI + ksub;
      kraw[u] = (j < len) ? ld_stream16(kc + ((int64_t)b * Smax + j) * HD + (int64_t)h * D + kpart * 8) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = j0 + u * KPI + ksub;
      float kv[8];
      unpack8<DT>(kraw[u], kv);
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s = fmaf(q8[e], kv[e], s);
#pragma unroll
      for (int o2 = OCT1 / 2; o2 > 0; o2 >>= 1) s += __shfl_xor(s, o2, 64);
      s *= scale_log2;
      if (j < len) {
        if (kpart == 0) sc[j] = s;
        mx = fmaxf(mx, s);
      }
    }
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int j = tid; j < len; j += 256) {
    const float pj = fast_exp2(sc[j] - mx);
    sc[j] = pj;
    sum += pj;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  sum = (red[4] + red[5]) + (red[6] + red[7]);
  // pass 2: thread = (key slice g of 256*8/D, channel octet c); 8 channels per thread
  constexpr int OCT = D / 8;       // octets per value row
  constexpr int G = 256 / OCT;     // key slices
  const int c = tid % OCT, gsl = tid / OCT;
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
  for (int j0 = gsl; j0 < len; j0 += G * UNR) {
    uint4 vraw[UNR];
    float pj[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = j0 + u * G;
      const bool ok = j < len;
      vraw[u] = ok ? ld_stream16(vc + ((int64_t)b * Smax + j) * HD + (int64_t)h * D + c * 8) : make_uint4(0, 0, 0, 0);
      pj[u] = ok ? sc[j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float vv[8];
      unpack8<DT>(vraw[u], vv);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = fmaf(pj[u], vv[e], o[e]);
    }
  }
  __syncthreads();  // everyone is done reading the scores: reuse the buffer for the slice partials
  float* part = sc;  // [G][D]
#pragma unroll
  for (int e = 0; e < 8; ++e) part[gsl * D + c * 8 + e] = o[e];
  __syncthreads();
  if (tid < D) {
    float a = 0.f;
    for (int g2 = 0; g2 < G; ++g2) a += part[g2 * D + tid];
    if (splits == 1) {
      out[(int64_t)b * HD + (int64_t)h * D + tid] = (uint16_t)st16<DT>(len > 0 ? a / sum : 0.f);
    } else {
      float* w = ws + ((int64_t)(b * H + h) * splits + sp) * (D + 2);
      if (cnt) {  // device-coherent stores: the merging block may sit on another XCD (its L2 is not this one's)
        st_agent(w + tid, a);
        if (tid == 0) { st_agent(w + D, mx); st_agent(w + D + 1, sum); }
      } else {
        w[tid] = a;
        if (tid == 0) { w[D] = mx; w[D + 1] = sum; }
      }
    }
  }
  if (splits == 1 || !cnt) return;
  // The last of the `splits` blocks of this (b, h) to get here merges the partial softmaxes: no second launch (at batch 1 the merge
  // kernel is 5.6 us of dependent round trips per layer - and so is this tail: no gain).  Ticket counter per (b, h), reset by its last taker.
  // (no __threadfence: at agent scope it writes back and invalidates the whole L2 - measured 150 us per layer.  The partials are
  //  written and read with device-coherent accesses, so all the release needs is that this block's stores have completed)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(cnt + (b * H + h), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = (t == (unsigned)splits - 1u);
    if (last) __hip_atomic_store(cnt + (b * H + h), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    red[0] = last ? 1.f : 0.f;
  }
  __syncthreads();
  if (red[0] == 0.f) return;
  asm volatile("" ::: "memory");
  if (tid < D) {
    const float* w = ws + (int64_t)(b * H + h) * splits * (D + 2);
    float v[32], sm_[32], sl_[32];
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) {  // every partial is requested before any is used
      v[s2] = s2 < splits ? ld_agent(w + s2 * (D + 2) + tid) : 0.f;
      sm_[s2] = s2 < splits ? ld_agent(w + s2 * (D + 2) + D) : -1e30f;
      sl_[s2] = s2 < splits ? ld_agent(w + s2 * (D + 2) + D + 1) : 0.f;
    }
    float M = -1e30f;
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) M = fmaxf(M, sm_[s2]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) {
      if (s2 < splits) {
        const float f = fast_exp2(sm_[s2] - M);
        num += v[s2] * f;
        den += sl_[s2] * f;
      }
    }
    out[(int64_t)b * HD + (int64_t)h * D + tid] = (uint16_t)st16<DT>(den > 0.f ? num / den : 0.f);
  }
}

// (splits <= 32: every partial is requested before any is used - the kernel is a handful of dependent L2 round trips otherwise)
template <int DT, int D>
__global__ __launch_bounds__(D) void attn_decode_combine_k(const float* __restrict__ ws, uint16_t* __restrict__ out, int H, int splits) {
  __shared__ float sm[32], sl[32];
  const int bh = blockIdx.x, tid = threadIdx.x;
  const float* w = ws + (int64_t)bh * splits * (D + 2);
  float v[32];
#pragma unroll
  for (int s2 = 0; s2 < 32; ++s2) v[s2] = s2 < splits ? w[s2 * (D + 2) + tid] : 0.f;
  if (tid < splits) {
    sm[tid] = w[tid * (D + 2) + D];
    sl[tid] = w[tid * (D + 2) + D + 1];
  }
  __syncthreads();
  float M = -1e30f;
  for (int s2 = 0; s2 < splits; ++s2) M = fmaxf(M, sm[s2]);
  float num = 0.f, den = 0.f;
#pragma unroll
  for (int s2 = 0; s2 < 32; ++s2) {
    if (s2 < splits) {
      const float f = fast_exp2(sm[s2] - M);
      num += v[s2] * f;
      den += sl[s2] * f;
    }
  }
  out[(int64_t)bh * D + tid] = (uint16_t)st16<DT>(den > 0.f ? num / den : 0.f);
}

}  // namespace

// activation-row count from which the MFMA form is used (measured, profiles/r02_gemv_ab.txt: 3 rows for both weight formats - below
// that the one-wave-per-row kernels stream faster); mh_gemv_mfma_min_rows(r) overrides both (A/B switch; 17 = never)
static int g_gemv_mfma_min_rows = 3, g_gemv_mfma_min_rows_fp8 = 3;
// the forms with a paired epilogue (SwiGLU, RoPE + append) compete with one-wave-per-row-pair kernels that already fuse the same work
// and stream at 4.3-4.5 TB/s up to ~5 rows: measured crossover 6 rows with 16-bit weights, 4 with fp8 (mh_gemv_mfma_pair_min_rows)
static int g_gemv_mfma_pair_min = 6, g_gemv_mfma_pair_min_fp8 = 4;
static int g_gemv_ksplit = 1;  // 1-2 rows, N <= 8192: K split over the four waves of a block (A-B switch: mh_gemv_ksplit)
extern "C" void mh_gemv_ksplit(int on) { g_gemv_ksplit = on ? 1 : 0; }
extern "C" void mh_gemv_mfma_min_rows(int rows) {
  if (rows <= 0) { g_gemv_mfma_min_rows = 3; g_gemv_mfma_min_rows_fp8 = 3; g_gemv_mfma_pair_min = 6; g_gemv_mfma_pair_min_fp8 = 4; }  // the defaults
  else g_gemv_mfma_min_rows = g_gemv_mfma_min_rows_fp8 = g_gemv_mfma_pair_min = g_gemv_mfma_pair_min_fp8 = rows;
}
extern "C" void mh_gemv_mfma_pair_min_rows(int rows16, int rows_fp8) {
  g_gemv_mfma_pair_min = rows16 > 0 ? rows16 : 6;
  g_gemv_mfma_pair_min_fp8 = rows_fp8 > 0 ? rows_fp8 : 4;
}

static int gemv_impl(const void* x, int64_t ldx, const void* W, int64_t ldw, void* out, int64_t ldo, const void* resid,
                     int64_t ldr, int M, int N, int K, int dt, int out_f32, int swi_ff, const void* norm_w, float eps, const RopeAppend& ra,
                     void* stream) {
  if (!x || !W || !out || M <= 0 || M > 16 || N <= 0 || K <= 0 || (K & 7) || (ldx & 7) || (ldw & 7)) return MH_ERR_ARG;
  if (!aligned16(x) || !aligned16(W)) return MH_ERR_ARG;
  if (dt != MH_BF16 && dt != MH_F16) return MH_ERR_DTYPE;
  if (norm_w && (M > 8 || K > 8192 || !aligned16(norm_w))) return MH_ERR_ARG;  // fused RMSNorm: row-per-wave form, whole row in LDS
  if (ra.tab && (swi_ff || (N & 1))) return MH_ERR_ARG;
  if ((swi_ff || ra.tab) && (resid || out_f32)) return MH_ERR_ARG;
  // 3+ rows: the MFMA form (above) is HBM-bound where this one turns VALU-bound (the rotary-pair form needs D/2 to be whole 16-row groups)
  if (!norm_w && M >= ((swi_ff || ra.tab) ? g_gemv_mfma_pair_min : g_gemv_mfma_min_rows) && (K % 32) == 0 &&
      (!ra.tab || ((ra.D >> 1) % 16 == 0 && N == 3 * ra.H * ra.D))) {
    if (dt == MH_BF16) return launch_gemv_mfma<MH_BF16, false>(x, ldx, W, ldw, nullptr, out, ldo, resid, ldr, M, N, K, out_f32, swi_ff, ra, as_stream(stream));
    return launch_gemv_mfma<MH_F16, false>(x, ldx, W, ldw, nullptr, out, ldo, resid, ldr, M, N, K, out_f32, swi_ff, ra, as_stream(stream));
  }
  if (M > 8) return MH_ERR_ARG;
  // weight rows per wave: as many as keep >= ~1000 blocks in flight (N = 4096 with 4 rows per wave is 256 blocks = one per
  // CU, measured at 1.4 TB/s; with 1 row per wave 3+ TB/s)
  if (g_gemv_ksplit && M <= 2 && !swi_ff && !norm_w && !ra.tab && N <= 8192 && K >= 2048) {  // small N, 1-2 rows: K split over the block's waves
    const dim3 gridk((N + 1) / 2), blockk(256);
    hipStream_t stk = as_stream(stream);
    if (dt == MH_BF16) {
      if (M == 1) hipLaunchKernelGGL((gemv_ks_k<MH_BF16, 1, 2>), gridk, blockk, 0, stk, (const uint16_t*)x, ldx, (const uint16_t*)W, ldw, out, ldo, (const uint16_t*)resid, ldr, N, K, out_f32);
      else hipLaunchKernelGGL((gemv_ks_k<MH_BF16, 2, 2>), gridk, blockk, 0, stk, (const uint16_t*)x, ldx, (const uint16_t*)W, ldw, out, ldo, (const uint16_t*)resid, ldr, N, K, out_f32);
    } else {
      if (M == 1) hipLaunchKernelGGL((gemv_ks_k<MH_F16, 1, 2>), gridk, blockk, 0, stk, (const uint16_t*)x, ldx, (const uint16_t*)W, ldw, out, ldo, (const uint16_t*)resid, ldr, N, K, out_f32);
      else hipLaunchKernelGGL((gemv_ks_k<MH_F16, 2, 2>), gridk, blockk, 0, stk, (const uint16_t*)x, ldx, (const uint16_t*)W, ldw, out, ldo, (const uint16_t*)resid, ldr, N, K, out_f32);
    }
    MH_LAUNCH_CHECK();
  }
  // (fused SwiGLU: a wave's rows are gate/up PAIRS, so an even count; N counts outputs = pairs)
  const int rows = ra.tab ? 2 : swi_ff ? (M < 3 ? 2 : 4) : (M < 3 ? 2 : (N >= 16384 ? 4 : (N >= 8192 ? 2 : 1)));
  const int cols = swi_ff ? rows / 2 : rows;  // output columns per wave
  const dim3 grid((N + 4 * cols - 1) / (4 * cols)), block(256);
  hipStream_t st = as_stream(stream);
#define GO1(DT_, MM_, R_, L_, NS_, NRM_, LDS_)                                                                                     \
  do {                                                                                                                             \
    const size_t lds_ = (LDS_);                                                                                                    \
    static bool attr_ = false;                                                                                                     \
    if (lds_ && !attr_) {                                                                                                          \
      hipFuncSetAttribute((const void*)gemv_k<DT_, MM_, R_, L_, NS_, NRM_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(MM_ * 8192 * 2)); \
      attr_ = true;                                                                                                                \
    }                                                                                                                              \
    hipLaunchKernelGGL((gemv_k<DT_, MM_, R_, L_, NS_, NRM_>), grid, block, lds_, st, (const uint16_t*)x, ldx, (const uint16_t*)W, ldw, out, ldo, \
                       (const uint16_t*)resid, ldr, N, K, out_f32, swi_ff, (const uint16_t*)norm_w, eps, ra);                      \
  } while (0)
#define GO(DT_, MM_, R_, L_, NS_)                                                                                                  \
  do {                                                                                                                             \
    if (norm_w) GO1(DT_, MM_, R_, true, NS_, true, (size_t)MM_ * K * 2);                                                           \
    else GO1(DT_, MM_, R_, L_, NS_, false, L_ ? (size_t)MM_ * GEMV_KC * 2 : 0);                                                    \
  } while (0)
#define GOR(DT_, MM_)                                                                  \
  do {                                                                                 \
    if (rows == 4) GO(DT_, MM_, 4, true, 2); else if (rows == 2) GO(DT_, MM_, 2, true, 2); else GO(DT_, MM_, 1, true, 2); \
  } while (0)
#define GOS(DT_, MM_) GO(DT_, MM_, 2, false, 2) /* 1-2 activation rows: no LDS staging */
#define GOM(DT_)                                                                                                       \
  switch (M) {                                                                                                         \
    case 1: GOS(DT_, 1); break; case 2: GOS(DT_, 2); break; case 3: GOR(DT_, 3); break;                               \
    case 4: GOR(DT_, 4); break; case 5: GOR(DT_, 5); break; case 6: GOR(DT_, 6); break;                               \
    case 7: GOR(DT_, 7); break; default: GOR(DT_, 8); break;                                                           \
  }
  if (dt == MH_BF16) { GOM(MH_BF16); } else { GOM(MH_F16); }
#undef GOM
#undef GOS
#undef GOR
#undef GO
#undef GO1
  MH_LAUNCH_CHECK();
}

extern "C" int mh_gemv(const void* x, int64_t ldx, const void* W, int64_t ldw, void* out, int64_t ldo, const void* resid,
                       int64_t ldr, int M, int N, int K, int dt, int out_f32, void* stream) {
  return gemv_impl(x, ldx, W, ldw, out, ldo, resid, ldr, M, N, K, dt, out_f32, 0, nullptr, 0.f, RopeAppend{}, stream);
}
// act[M, ff] = silu(x Wg^T) * (x Wu^T) with Wgu = [Wg; Wu] [2 ff, K] (HF LlamaMLP gate / up of the decode step): one launch, the
// gate|up projection never reaches memory (gate / up are rounded to 16 bits before the activation, as the two launches do).  M <= 16 rows (MFMA form from 3 rows on).
extern "C" int mh_gemv_swiglu(const void* x, int64_t ldx, const void* Wgu, int64_t ldw, void* act, int64_t ldo, int M, int ff, int K, int dt,
                              void* stream) {
  if (ff <= 0) return MH_ERR_ARG;
  return gemv_impl(x, ldx, Wgu, ldw, act, ldo, nullptr, 0, M, ff, K, dt, 0, ff, nullptr, 0.f, RopeAppend{}, stream);
}
// The same two projections with the RMSNorm that precedes them (HF LlamaDecoderLayer: input_layernorm -> q|k|v, post_attention_layernorm ->
// gate|up) applied by the GEMV blocks themselves: y = rmsnorm(x; norm_w, eps) W^T, and (ff > 0) act = silu(.) * (.) of the gate|up rows.
// Equal to mh_rmsnorm_fwd + mh_gemv (+ mh_swiglu_fwd) bit for bit in the projection.  M <= 8, K <= 8192.
extern "C" int mh_gemv_norm(const void* x, int64_t ldx, const void* norm_w, float eps, const void* W, int64_t ldw, void* out, int64_t ldo, int M, int N,
                            int K, int ff, int dt, void* stream) {
  if (!norm_w || M > 8 || ff < 0) return MH_ERR_ARG;
  return gemv_impl(x, ldx, W, ldw, out, ldo, nullptr, 0, M, ff > 0 ? ff : N, K, dt, 0, ff, norm_w, eps, RopeAppend{}, stream);
}


extern "C" int mh_decode_rope_append(void* qkv, const float* cos_sin, const int32_t* pos, const int32_t* rope_pos, void* kcache, void* vcache,
                                     int B, int H, int D, int Smax, int dt, void* stream) {
  if (!rope_pos) rope_pos = pos;
  if (!qkv || !cos_sin || !pos || !kcache || !vcache || B <= 0 || H <= 0 || (D & 15) || Smax <= 0) return MH_ERR_ARG;
  if (dt != MH_BF16 && dt != MH_F16) return MH_ERR_DTYPE;
  const int64_t total = (int64_t)B * H * (D / 16);
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dt == MH_BF16)
    hipLaunchKernelGGL(rope_append_k<MH_BF16>, grid, block, 0, as_stream(stream), (uint16_t*)qkv, (const float2*)cos_sin, pos, rope_pos,
                       (uint16_t*)kcache, (uint16_t*)vcache, B, H, D, Smax);
  else
    hipLaunchKernelGGL(rope_append_k<MH_F16>, grid, block, 0, as_stream(stream), (uint16_t*)qkv, (const float2*)cos_sin, pos, rope_pos,
                       (uint16_t*)kcache, (uint16_t*)vcache, B, H, D, Smax);
  MH_LAUNCH_CHECK();
}

static int g_decode_fused_merge = 0;  // 1: split-KV merge by the last block of a (b, h) instead of a second launch (A-B arm: measured 0.3-1 % slower, profiles/r03_decode_nt_ab.txt)
extern "C" void mh_attn_decode_fused_merge(int on) { g_decode_fused_merge = on ? 1 : 0; }
extern "C" int mh_attn_decode_splits(int B, int H, int Smax) {
  int s = (1024 + B * H - 1) / (B * H);
  const int by_len = (Smax + 127) / 128;  // >= 128 keys per split
  if (s > by_len) s = by_len;
  if (s > 32) s = 32;
  return s < 1 ? 1 : s;
}

extern "C" int mh_attn_decode(const void* q, int64_t ldq, const void* kcache, const void* vcache, void* out, const int32_t* lens,
                              int B, int H, int D, int Smax, float* ws, int dt, void* stream) {
  if (!q || !kcache || !vcache || !out || !lens || B <= 0 || H <= 0 || Smax <= 0 || (ldq & 7)) return MH_ERR_ARG;
  if (D != 128 && D != 64) return MH_ERR_SHAPE;
  if (dt != MH_BF16 && dt != MH_F16) return MH_ERR_DTYPE;
  const float scale_log2 = 1.4426950408889634f / sqrtf((float)D);
  const int splits = ws ? mh_attn_decode_splits(B, H, Smax) : 1;
  const int chunk = (Smax + splits - 1) / splits;
  const int G = 256 / (D / 8);
  const size_t lds = sizeof(float) * (size_t)((chunk > G * D) ? chunk : G * D);
  if (lds > 150 * 1024) return MH_ERR_SHAPE;  // <= 38400 keys per split
  const dim3 grid(B * H * splits), block(256);
  hipStream_t st = as_stream(stream);
  unsigned* cnt = nullptr;
  if (splits > 1 && g_decode_fused_merge && B * H <= 16384) {
    static unsigned* tickets = nullptr;
    if (!tickets && hipGetSymbolAddress((void**)&tickets, HIP_SYMBOL(g_decode_tickets)) != hipSuccess) tickets = nullptr;
    cnt = tickets;
  }
#define GO(DT_, D_)                                                                                                     \
  do {                                                                                                                   \
    static bool attr = false;                                                                                           \
    if (!attr) {                                                                                                         \
      hipFuncSetAttribute((const void*)attn_decode_k<DT_, D_>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
      attr = true;                                                                                                       \
    }                                                                                                                    \
    hipLaunchKernelGGL((attn_decode_k<DT_, D_>), grid, block, lds, st, (const uint16_t*)q, ldq, (const uint16_t*)kcache, \
                       (const uint16_t*)vcache, (uint16_t*)out, lens, H, Smax, scale_log2, splits, chunk, ws, cnt);      \
    if (splits > 1 && !cnt)                                                                                              \
      hipLaunchKernelGGL((attn_decode_combine_k<DT_, D_>), dim3(B * H), dim3(D_), 0, st, (const float*)ws, (uint16_t*)out, H, splits); \
  } while (0)
  if (dt == MH_BF16) { if (D == 128) GO(MH_BF16, 128); else GO(MH_BF16, 64); }
  else { if (D == 128) GO(MH_F16, 128); else GO(MH_F16, 64); }
#undef GO
  MH_LAUNCH_CHECK();
}

extern "C" int mh_quant_fp8_b128(const void* w, int64_t ldw, void* q, float* scales, int N, int K, int dt, void* stream) {
  if (!w || !q || !scales || N <= 0 || K <= 0 || (K & 7) || (ldw & 7) || !aligned16(w)) return MH_ERR_ARG;
  if (dt != MH_BF16 && dt != MH_F16) return MH_ERR_DTYPE;
  const int64_t blocks = (int64_t)N * ((K + 127) / 128);
  const dim3 grid((unsigned)((blocks + 15) / 16)), block(256);
  if (dt == MH_BF16)
    hipLaunchKernelGGL(quant_fp8_b128_k<MH_BF16>, grid, block, 0, as_stream(stream), (const uint16_t*)w, ldw, (uint8_t*)q, scales, N, K);
  else
    hipLaunchKernelGGL(quant_fp8_b128_k<MH_F16>, grid, block, 0, as_stream(stream), (const uint16_t*)w, ldw, (uint8_t*)q, scales, N, K);
  MH_LAUNCH_CHECK();
}

static int gemv_fp8w_impl(const void* x, int64_t ldx, const void* q, const float* scales, void* out, int64_t ldo, const void* resid,
                          int64_t ldr, int M, int N, int K, int dt, int out_f32, int swi_ff, const void* norm_w, float eps, const RopeAppend& ra,
                          void* stream) {
  if (!x || !q || !scales || !out || M <= 0 || M > 16 || N <= 0 || K <= 0 || (K & 15) || (ldx & 7)) return MH_ERR_ARG;
  if (!aligned16(x) || !aligned16(q)) return MH_ERR_ARG;
  if (dt != MH_BF16 && dt != MH_F16) return MH_ERR_DTYPE;
  if (norm_w && (M > 8 || K > 8192 || !aligned16(norm_w))) return MH_ERR_ARG;
  if (ra.tab && (swi_ff || (N & 1))) return MH_ERR_ARG;
  if ((swi_ff || ra.tab) && (resid || out_f32)) return MH_ERR_ARG;
  if (!norm_w && M >= ((swi_ff || ra.tab) ? g_gemv_mfma_pair_min_fp8 : g_gemv_mfma_min_rows_fp8) && (K % 64) == 0 &&
      (!ra.tab || ((ra.D >> 1) % 16 == 0 && N == 3 * ra.H * ra.D))) {
    if (dt == MH_BF16) return launch_gemv_mfma<MH_BF16, true>(x, ldx, q, K, scales, out, ldo, resid, ldr, M, N, K, out_f32, swi_ff, ra, as_stream(stream));
    return launch_gemv_mfma<MH_F16, true>(x, ldx, q, K, scales, out, ldo, resid, ldr, M, N, K, out_f32, swi_ff, ra, as_stream(stream));
  }
  if (M > 8 || ((swi_ff || ra.tab) && K > 8192)) return MH_ERR_ARG;
  if (g_gemv_ksplit && M <= 2 && !swi_ff && !norm_w && !ra.tab && N <= 8192 && K >= 2048) {  // small N, 1-2 rows: K split over the block's waves
    const dim3 gridk((N + 1) / 2), blockk(256);
    hipStream_t stk = as_stream(stream);
    if (dt == MH_BF16) {
      if (M == 1) hipLaunchKernelGGL((gemv_fp8w_ks_k<MH_BF16, 1, 2>), gridk, blockk, 0, stk, (const uint16_t*)x, ldx, (const uint8_t*)q, scales, out, ldo, (const uint16_t*)resid, ldr, N, K, out_f32);
      else hipLaunchKernelGGL((gemv_fp8w_ks_k<MH_BF16, 2, 2>), gridk, blockk, 0, stk, (const uint16_t*)x, ldx, (const uint8_t*)q, scales, out, ldo, (const uint16_t*)resid, ldr, N, K, out_f32);
    } else {
      if (M == 1) hipLaunchKernelGGL((gemv_fp8w_ks_k<MH_F16, 1, 2>), gridk, blockk, 0, stk, (const uint16_t*)x, ldx, (const uint8_t*)q, scales, out, ldo, (const uint16_t*)resid, ldr, N, K, out_f32);
      else hipLaunchKernelGGL((gemv_fp8w_ks_k<MH_F16, 2, 2>), gridk, blockk, 0, stk, (const uint16_t*)x, ldx, (const uint8_t*)q, scales, out, ldo, (const uint16_t*)resid, ldr, N, K, out_f32);
    }
    MH_LAUNCH_CHECK();
  }
  const int rows = (swi_ff || ra.tab) ? 2 : (M < 3 ? 1 : (N >= 8192 ? 2 : 1));  // weight rows per wave (>= ~1000 blocks in flight, as in mh_gemv); SwiGLU: one gate/up pair
  const int cols = swi_ff ? rows / 2 : rows;
  const dim3 grid((N + 4 * cols - 1) / (4 * cols)), block(256);
  hipStream_t st = as_stream(stream);
#define GO1(DT_, MM_, R_, L_, NRM_, LDS_)                                                                                           \
  do {                                                                                                                              \
    const size_t lds_ = (LDS_);                                                                                                     \
    static bool attr_ = false;                                                                                                      \
    if (lds_ && !attr_) {                                                                                                           \
      hipFuncSetAttribute((const void*)gemv_fp8w_k<DT_, MM_, R_, L_, NRM_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(MM_ * 8192 * 2)); \
      attr_ = true;                                                                                                                 \
    }                                                                                                                               \
    hipLaunchKernelGGL((gemv_fp8w_k<DT_, MM_, R_, L_, NRM_>), grid, block, lds_, st, (const uint16_t*)x, ldx, (const uint8_t*)q, scales, \
                       out, ldo, (const uint16_t*)resid, ldr, N, K, out_f32, swi_ff, (const uint16_t*)norm_w, eps, ra);            \
  } while (0)
#define GO(DT_, MM_, R_, L_)                                                                                                       \
  do {                                                                                                                              \
    if (norm_w) GO1(DT_, MM_, R_, true, true, (size_t)MM_ * K * 2);                                                                 \
    else GO1(DT_, MM_, R_, L_, false, L_ ? (size_t)MM_ * GEMV_KC * 2 : 0);                                                          \
  } while (0)
#define GOR(DT_, MM_)                                                                        \
  do {                                                                                       \
    if (rows == 2) GO(DT_, MM_, 2, true); else GO(DT_, MM_, 1, true);                       \
  } while (0)
#define GOS(DT_, MM_)                                                                        \
  do {                                                                                       \
    if (rows == 2) GO(DT_, MM_, 2, false); else GO(DT_, MM_, 1, false);                     \
  } while (0)
#define GOM(DT_)                                                                                                   \
  switch (M) {                                                                                                     \
    case 1: GOS(DT_, 1); break; case 2: GOS(DT_, 2); break; case 3: GOR(DT_, 3); break;                           \
    case 4: GOR(DT_, 4); break; case 5: GOR(DT_, 5); break; case 6: GOR(DT_, 6); break;                           \
    case 7: GOR(DT_, 7); break; default: GOR(DT_, 8); break;                                                       \
  }
  if (dt == MH_BF16) { GOM(MH_BF16); } else { GOM(MH_F16); }
#undef GOM
#undef GOS
#undef GOR
#undef GO
#undef GO1
  MH_LAUNCH_CHECK();
}

extern "C" int mh_gemv_fp8w(const void* x, int64_t ldx, const void* q, const float* scales, void* out, int64_t ldo, const void* resid,
                            int64_t ldr, int M, int N, int K, int dt, int out_f32, void* stream) {
  return gemv_fp8w_impl(x, ldx, q, scales, out, ldo, resid, ldr, M, N, K, dt, out_f32, 0, nullptr, 0.f, RopeAppend{}, stream);
}
// mh_gemv_norm with fp8 (e4m3, per-128-block scales) weights: out = rmsnorm(x; norm_w, eps) W^T (norm_w may be NULL: no norm), ff > 0: SwiGLU of the
// gate|up rows.  With norm_w: M <= 8, K <= 8192; without: M <= 16.
extern "C" int mh_gemv_fp8w_norm(const void* x, int64_t ldx, const void* norm_w, float eps, const void* q, const float* scales, void* out,
                                 int64_t ldo, int M, int N, int K, int ff, int dt, void* stream) {
  if (ff < 0) return MH_ERR_ARG;
  return gemv_fp8w_impl(x, ldx, q, scales, out, ldo, nullptr, 0, M, ff > 0 ? ff : N, K, dt, 0, ff, norm_w, eps, RopeAppend{}, stream);
}

// q|k|v projection of the decode step with everything around it in one launch: (optional) input_layernorm of x, the projection with 16-bit
// (W) or fp8 (q8 + scales) weights, rotate-half RoPE of q and k at pos[m] and the append of k, v to the cache rows [m, pos[m]]
// (= mh_rmsnorm_fwd + mh_gemv / mh_gemv_fp8w + mh_decode_rope_append, bit for bit).  qkv [M, 3 H D] receives the rotated q, k and v.
extern "C" int mh_gemv_qkv_rope(const void* x, int64_t ldx, const void* norm_w, float eps, const void* W, int64_t ldw, const void* q8,
                                const float* scales, void* qkv, int64_t ldo, int M, int K, int dt, const float* cos_sin, const int32_t* pos,
                                const int32_t* rope_pos, void* kcache, void* vcache, int H, int D, int Smax, void* stream) {
  if (!cos_sin || !pos || !kcache || !vcache || H <= 0 || D <= 0 || (D & 1) || Smax <= 0 || (!W && !(q8 && scales))) return MH_ERR_ARG;
  RopeAppend ra;
  ra.tab = (const float2*)cos_sin; ra.pos = pos; ra.rpos = rope_pos ? rope_pos : pos; ra.kc = (uint16_t*)kcache; ra.vc = (uint16_t*)vcache; ra.H = H; ra.D = D; ra.Smax = Smax;
  const int N = 3 * H * D;
  if (W) return gemv_impl(x, ldx, W, ldw, qkv, ldo, nullptr, 0, M, N, K, dt, 0, 0, norm_w, eps, ra, stream);
  return gemv_fp8w_impl(x, ldx, q8, scales, qkv, ldo, nullptr, 0, M, N, K, dt, 0, 0, norm_w, eps, ra, stream);
}
