// fp8 (OCP e4m3) operand preparation for the scaled-fp8 MFMA GEMMs of the training step (BASELINE cfg 5's weight path).
//
// Every GEMM of the fp8 step is an NT product of two ROW-quantised operands (one fp32 scale per row, contraction along the
// row): forward x W^T uses rowquant(x), rowquant(W); dgrad dy W = dy (W^T)^T uses rowquant(dy), rowquant(W^T); wgrad
// dy^T x uses rowquant(dy^T), rowquant(x^T) with the contraction over tokens.  So next to mh_quant_fp8_rows (gemm.hip) the step
// needs the TRANSPOSED form: q_t[c, r] = e4m3(x[r, c] / s[c]), s[c] = max_r |x[r, c]| / 448 - a column-scaled, transposed,
// zero-padded copy, produced here in two HBM-bound passes (column maxima, then a transpose through LDS with 4 rows packed
// per 32-bit LDS store and 32-byte row-major writes).  5 bytes of traffic per element (2 x 2 read + 1 written).
#include "mh_common.h"

namespace {

// amax[c] (uint bits of a non-negative float; order-preserving) = max over rows of |x[r, c]|
template <int DT>
__global__ __launch_bounds__(256) void col_absmax_k(const uint16_t* __restrict__ x, int64_t ldx, unsigned* __restrict__ amax, int R, int C) {
  const int cv = blockIdx.x * 256 + threadIdx.x;  // 8-column group
  if (cv * 8 >= C) return;
  const int rows_per = (R + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
  float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // eight row pieces are requested before any is used (clamped addresses; a repeated last row does not change a maximum): one dependent
  // round trip per row held this pass at 1.6 TB/s (profiles/r05_step_cfg5_fp8_kernel_stats.txt)
  const uint16_t* xc = x + cv * 8;
  for (int r = r0; r < r1; r += 8) {
    uint4 raw[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) raw[u] = *(const uint4*)(xc + (int64_t)min(r + u, r1 - 1) * ldx);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float f[8];
      unpack8<DT>(raw[u], f);
#pragma unroll
      for (int i = 0; i < 8; ++i) m[i] = fmaxf(m[i], fabsf(f[i]));
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (m[i] > 0.f) atomicMax(amax + cv * 8 + i, __float_as_uint(m[i]));
}

__global__ __launch_bounds__(256) void amax_to_scale_k(const unsigned* __restrict__ amax, float* __restrict__ sc, int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < C) {
    const float m = __uint_as_float(amax[i]);
    sc[i] = m > 0.f ? m * (1.0f / 448.0f) : 1.0f;
  }
}

// tile: 64 columns x 128 rows -> q_t[64 rows of the output][128 bytes]
template <int DT>
__global__ __launch_bounds__(256) void quant_fp8_transposed_k(const uint16_t* __restrict__ x, int64_t ldx, const float* __restrict__ sc,
                                                              uint8_t* __restrict__ qt, int64_t ldq, int R, int C) {
  __shared__ unsigned tile[64][33];  // [column][32 row-quads] (+1: conflict-free column-major writes)
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 128;
  const int cg = threadIdx.x & 7, rq = threadIdx.x >> 3;
  const int col = c0 + cg * 8;
  // (the row pieces are requested first, unconditionally at clamped addresses, and masked below: inside the bounds test they went out one
  //  dependent round trip at a time)
  uint4 raw[4];
  const int colc = min(col, C - 8);
#pragma unroll
  for (int i = 0; i < 4; ++i) raw[i] = *(const uint4*)(x + (int64_t)min(r0 + rq * 4 + i, R - 1) * ldx + colc);
  float inv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) inv[j] = (col + j < C) ? 1.0f / sc[colc + j] : 0.f;
  float v[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + rq * 4 + i;
    if (r < R && col < C) {
      unpack8<DT>(raw[i], v[i]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(v[0][j] * inv[j], v[1][j] * inv[j], 0, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(v[2][j] * inv[j], v[3][j] * inv[j], p, true);
    tile[cg * 8 + j][rq] = (unsigned)p;
  }
  __syncthreads();
  // write-out: 64 output rows x 128 B; thread -> (output row = t >> 2, 32-byte segment = t & 3)
  const int orow = threadIdx.x >> 2, seg = threadIdx.x & 3;
  if (c0 + orow < C) {
    uint4 a, b;
    a.x = tile[orow][seg * 8 + 0]; a.y = tile[orow][seg * 8 + 1]; a.z = tile[orow][seg * 8 + 2]; a.w = tile[orow][seg * 8 + 3];
    b.x = tile[orow][seg * 8 + 4]; b.y = tile[orow][seg * 8 + 5]; b.z = tile[orow][seg * 8 + 6]; b.w = tile[orow][seg * 8 + 7];
    uint4* dst = (uint4*)(qt + (int64_t)(c0 + orow) * ldq + r0 + seg * 32);
    dst[0] = a;
    dst[1] = b;
  }
}

// ---- weights: per-row fp32 scale + per-(row, 128-k block) 4-bit exponent e (block scale = s[n] * 2^-e), the form the fp8 GEMM applies
// through the MFMA's E8M0 block-scale operand (BASELINE cfg 5: "per-128-block scales").  e = floor(log2(rowmax / blockmax)) in [0, 15]:
// every block is scaled up by the largest power of two that keeps it inside the row's range, so small blocks keep e4m3's full
// 3-bit mantissa instead of sinking towards the subnormals.  Exponent image: [row / 128][G][...] as documented in merlin_hip.h.
__device__ __forceinline__ int block_exp(float rowmax, float bmax) {
  if (!(bmax > 0.f)) return 15;
  int ea, eb;
  const float ma = frexpf(rowmax, &ea), mb = frexpf(bmax, &eb);
  const int e = ea - eb - (ma < mb ? 1 : 0);  // exact floor(log2(rowmax / bmax))
  return min(15, max(0, e));
}

// one wave per PAIR of rows (2p, 2p + 1): the two rows share every exponent byte
template <int DT>
__global__ __launch_bounds__(256) void quant_fp8_rows_e4_k(const uint16_t* __restrict__ w, int64_t ldw, uint8_t* __restrict__ q, float* __restrict__ sc,
                                                           uint8_t* __restrict__ exh, int64_t G, int N, int K) {
  int* flag = (int*)exh;          // 16-byte header: "some exponent is non-zero"
  uint8_t* ex = exh + 16;
  const int lane = threadIdx.x & 63;
  const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
  if (n0 >= N) return;
  const bool two = n0 + 1 < N;
  const uint4* r0 = (const uint4*)(w + (int64_t)n0 * ldw);
  const uint4* r1 = (const uint4*)(w + (int64_t)(two ? n0 + 1 : n0) * ldw);
  const int nch = K >> 3;  // K % 128 == 0: 16 chunks per block
  float m0 = 0.f, m1 = 0.f;
  for (int c = lane; c < nch; c += 64) {
    float f[8];
    unpack8<DT>(r0[c], f);
#pragma unroll
    for (int i = 0; i < 8; ++i) m0 = fmaxf(m0, fabsf(f[i]));
    unpack8<DT>(r1[c], f);
#pragma unroll
    for (int i = 0; i < 8; ++i) m1 = fmaxf(m1, fabsf(f[i]));
  }
  m0 = wave_max(m0);
  m1 = wave_max(m1);
  const float s0 = m0 > 0.f ? m0 * (1.0f / 448.0f) : 1.0f, s1 = m1 > 0.f ? m1 * (1.0f / 448.0f) : 1.0f;
  if (lane == 0) {
    sc[n0] = s0;
    if (two) sc[n0 + 1] = s1;
  }
  uint8_t* eb = ex + (int64_t)(n0 >> 7) * G + ((n0 & 127) >> 1);
  for (int c = lane; c < nch; c += 64) {
    float f0[8], f1[8];
    unpack8<DT>(r0[c], f0);
    unpack8<DT>(r1[c], f1);
    float b0 = 0.f, b1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { b0 = fmaxf(b0, fabsf(f0[i])); b1 = fmaxf(b1, fabsf(f1[i])); }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { b0 = fmaxf(b0, __shfl_xor(b0, o, 64)); b1 = fmaxf(b1, __shfl_xor(b1, o, 64)); }
    const int e0 = block_exp(m0, b0), e1 = two ? block_exp(m1, b1) : 0;
    const float i0 = ldexpf(1.0f / s0, e0), i1 = ldexpf(1.0f / s1, e1);
    const int kb = c >> 4;
    if ((lane & 15) == 0) {
      eb[(int64_t)kb * 64] = (uint8_t)(e0 | (e1 << 4));
      if ((e0 | e1) != 0 && *flag == 0) atomicOr(flag, 1);
    }
    int p0 = __builtin_amdgcn_cvt_pk_fp8_f32(f0[0] * i0, f0[1] * i0, 0, false);
    p0 = __builtin_amdgcn_cvt_pk_fp8_f32(f0[2] * i0, f0[3] * i0, p0, true);
    int p1 = __builtin_amdgcn_cvt_pk_fp8_f32(f0[4] * i0, f0[5] * i0, 0, false);
    p1 = __builtin_amdgcn_cvt_pk_fp8_f32(f0[6] * i0, f0[7] * i0, p1, true);
    *(uint2*)(q + (int64_t)n0 * K + c * 8) = make_uint2((unsigned)p0, (unsigned)p1);
    if (two) {
      p0 = __builtin_amdgcn_cvt_pk_fp8_f32(f1[0] * i1, f1[1] * i1, 0, false);
      p0 = __builtin_amdgcn_cvt_pk_fp8_f32(f1[2] * i1, f1[3] * i1, p0, true);
      p1 = __builtin_amdgcn_cvt_pk_fp8_f32(f1[4] * i1, f1[5] * i1, 0, false);
      p1 = __builtin_amdgcn_cvt_pk_fp8_f32(f1[6] * i1, f1[7] * i1, p1, true);
      *(uint2*)(q + (int64_t)(n0 + 1) * K + c * 8) = make_uint2((unsigned)p0, (unsigned)p1);
    }
  }
}

// transposed form (rows of the output = columns c of x, contraction along x's rows): a 64 x 128 tile holds exactly one 128-block of
// every output row it touches, so the block maximum is a reduction inside the tile
template <int DT>
__global__ __launch_bounds__(256) void quant_fp8_transposed_e4_k(const uint16_t* __restrict__ x, int64_t ldx, const float* __restrict__ sc,
                                                                 uint8_t* __restrict__ qt, int64_t ldq, uint8_t* __restrict__ exh, int64_t G, int R, int C) {
  int* flag = (int*)exh;
  uint8_t* ex = exh + 16;
  __shared__ unsigned tile[64][33];
  __shared__ float wmax[4][64];
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 128;
  const int cg = threadIdx.x & 7, rq = threadIdx.x >> 3, wave = threadIdx.x >> 6;
  const int col = c0 + cg * 8;
  float v[4][8], bm[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bm[j] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + rq * 4 + i;
    if (r < R && col < C) {
      unpack8<DT>(*(const uint4*)(x + (int64_t)r * ldx + col), v[i]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) bm[j] = fmaxf(bm[j], fabsf(v[i][j]));
  }
  // lanes with the same cg inside a wave: lane = 8 * (rq & 7) + cg
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) bm[j] = fmaxf(bm[j], __shfl_xor(bm[j], o, 64));
  if ((threadIdx.x & 63) < 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) wmax[wave][cg * 8 + j] = bm[j];
  }
  __syncthreads();
  unsigned ebytes = 0;
  float inv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float b = fmaxf(fmaxf(wmax[0][cg * 8 + j], wmax[1][cg * 8 + j]), fmaxf(wmax[2][cg * 8 + j], wmax[3][cg * 8 + j]));
    const float s = (col + j < C) ? sc[col + j] : 1.0f;
    const int e = block_exp(s * 448.0f, b);
    inv[j] = ldexpf(1.0f / s, e);
    ebytes |= (unsigned)e << (4 * j);  // nibble j: columns col + j, i.e. byte j/2, low nibble = even column
  }
  if (rq == 0 && col < C) {
    *(unsigned*)(ex + (int64_t)(col >> 7) * G + (int64_t)blockIdx.y * 64 + ((col & 127) >> 1)) = ebytes;
    if (ebytes != 0 && *flag == 0) atomicOr(flag, 1);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(v[0][j] * inv[j], v[1][j] * inv[j], 0, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(v[2][j] * inv[j], v[3][j] * inv[j], p, true);
    tile[cg * 8 + j][rq] = (unsigned)p;
  }
  __syncthreads();
  const int orow = threadIdx.x >> 2, seg = threadIdx.x & 3;
  if (c0 + orow < C) {
    uint4 a, b;
    a.x = tile[orow][seg * 8 + 0]; a.y = tile[orow][seg * 8 + 1]; a.z = tile[orow][seg * 8 + 2]; a.w = tile[orow][seg * 8 + 3];
    b.x = tile[orow][seg * 8 + 4]; b.y = tile[orow][seg * 8 + 5]; b.z = tile[orow][seg * 8 + 6]; b.w = tile[orow][seg * 8 + 7];
    uint4* dst = (uint4*)(qt + (int64_t)(c0 + orow) * ldq + r0 + seg * 32);
    dst[0] = a;
    dst[1] = b;
  }
}

// ---- row AND column maxima of x [R, C] in ONE read: 128 x 128 tiles, fp32-bit atomicMax (non-negative floats order like uints;
// max is order-independent, so the result is deterministic) --------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void absmax_rc_k(const uint16_t* __restrict__ x, int64_t ldx, unsigned* __restrict__ rmax, unsigned* __restrict__ cmax,
                                                   int R, int C) {
  __shared__ float colred[16][128];
  const int c0 = blockIdx.x * 128, r0 = blockIdx.y * 128;
  const int cg = threadIdx.x & 15, rq = threadIdx.x >> 4;  // 16 column groups of 8, 16 row groups of 8 rows
  const int col = c0 + cg * 8;
  float cm[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // all eight row pieces are requested before any is used (clamped addresses, masked below): inside the bounds test the loads were issued one
  // dependent round trip at a time and the pass ran at 2.9 TB/s
  uint4 raw[8];
  const int colc = min(col, C - 8);
#pragma unroll
  for (int i = 0; i < 8; ++i) raw[i] = *(const uint4*)(x + (int64_t)min(r0 + rq * 8 + i, R - 1) * ldx + colc);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = r0 + rq * 8 + i;
    float rm = 0.f;
    if (r < R && col < C) {
      float f[8];
      unpack8<DT>(raw[i], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float a = fabsf(f[j]);
        rm = fmaxf(rm, a);
        cm[j] = fmaxf(cm[j], a);
      }
    }
    // row maximum over the tile's 128 columns: the 16 lanes cg = 0..15 of one rq are consecutive lanes
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) rm = fmaxf(rm, __shfl_xor(rm, o, 64));
    if (cg == 0 && r < R && rm > 0.f) atomicMax(rmax + r, __float_as_uint(rm));
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) colred[rq][cg * 8 + j] = cm[j];
  __syncthreads();
  if (threadIdx.x < 128 && c0 + threadIdx.x < C) {
    float m = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) m = fmaxf(m, colred[q][threadIdx.x]);
    if (m > 0.f) atomicMax(cmax + c0 + threadIdx.x, __float_as_uint(m));
  }
}

// One read of a 128 x 128 tile -> its row-quantised bytes q[r, c] (scale sr[r]) AND its transposed column-quantised bytes
// qt[c, r] (scale sc[c]); both written as 128-byte row segments.
template <int DT>
__global__ __launch_bounds__(256) void quant_fp8_both_k(const uint16_t* __restrict__ x, int64_t ldx, const unsigned* __restrict__ rmax,
                                                        const unsigned* __restrict__ cmax, uint8_t* __restrict__ q, int64_t ldq,
                                                        float* __restrict__ sr, uint8_t* __restrict__ qt, int64_t ldqt, float* __restrict__ sc,
                                                        int R, int C) {
  __shared__ unsigned tile[128][33];  // [column][32 row-quads]
  const int c0 = blockIdx.x * 128, r0 = blockIdx.y * 128;
  const int cg = threadIdx.x & 15, rq = threadIdx.x >> 4;
  const int col = c0 + cg * 8;
  float cinv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float m = (col + j < C) ? __uint_as_float(cmax[col + j]) : 0.f;
    const float s = m > 0.f ? m * (1.0f / 448.0f) : 1.0f;
    cinv[j] = 1.0f / s;
    if (blockIdx.y == 0 && rq == 0 && col + j < C) sc[col + j] = s;
  }
  // the tile's eight row pieces and row maxima of this thread are requested up front (clamped addresses, masked below)
  uint4 raw[2][4];
  float rmx[2][4];
  const int colc = min(col, C - 8);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = min(r0 + pass * 64 + rq * 4 + i, R - 1);
      raw[pass][i] = *(const uint4*)(x + (int64_t)r * ldx + colc);
      rmx[pass][i] = __uint_as_float(rmax[r]);
    }
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float v[4][8];
    const int rb = r0 + pass * 64 + rq * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = rb + i;
      if (r < R && col < C) {
        unpack8<DT>(raw[pass][i], v[i]);
        const float m = rmx[pass][i];
        const float s = m > 0.f ? m * (1.0f / 448.0f) : 1.0f;
        const float inv = 1.0f / s;
        if (blockIdx.x == 0 && cg == 0) sr[r] = s;
        int p0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][0] * inv, v[i][1] * inv, 0, false);
        p0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][2] * inv, v[i][3] * inv, p0, true);
        int p1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][4] * inv, v[i][5] * inv, 0, false);
        p1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][6] * inv, v[i][7] * inv, p1, true);
        *(uint2*)(q + (int64_t)r * ldq + col) = make_uint2((unsigned)p0, (unsigned)p1);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int p = __builtin_amdgcn_cvt_pk_fp8_f32(v[0][j] * cinv[j], v[1][j] * cinv[j], 0, false);
      p = __builtin_amdgcn_cvt_pk_fp8_f32(v[2][j] * cinv[j], v[3][j] * cinv[j], p, true);
      tile[cg * 8 + j][pass * 16 + rq] = (unsigned)p;
    }
  }
  __syncthreads();
  // 128 output rows (columns of x) x 128 B: thread -> (row = t >> 1, 64-byte half = t & 1)
  const int orow = threadIdx.x >> 1, hf = threadIdx.x & 1;
  if (c0 + orow < C) {
    uint4* dst = (uint4*)(qt + (int64_t)(c0 + orow) * ldqt + r0 + hf * 64);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint4 a;
      a.x = tile[orow][hf * 16 + 4 * k + 0]; a.y = tile[orow][hf * 16 + 4 * k + 1];
      a.z = tile[orow][hf * 16 + 4 * k + 2]; a.w = tile[orow][hf * 16 + 4 * k + 3];
      dst[k] = a;
    }
  }
}

// out[0..m) = max(1e-30.., max_i s[i]): the tensor-wide scale of a row-quantised tensor = the largest of its row scales
__global__ __launch_bounds__(1024) void max_to_vec_k(const float* __restrict__ s, int n, float* __restrict__ out, int m) {
  __shared__ float red[16];
  float v = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) v = fmaxf(v, s[i]);
  v = wave_max(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) t = fmaxf(t, red[i]);
  if (!(t > 0.f)) t = 1.0f;
  for (int i = threadIdx.x; i < m; i += 1024) out[i] = t;
}

}  // namespace

// Transposed e4m3 copy with ONE scale for the whole tensor (scales[0]; the caller fills the [C] vector the GEMM reads with
// it: mh_max_to_vec): a single pass, 3 bytes of traffic per element.  e4m3 keeps its 3-bit mantissa over 2^15 of range, so
// a tensor-wide scale costs precision only for columns more than ~4 decades below the tensor maximum.
extern "C" int mh_quant_fp8_t_scaled(const void* x, int64_t ldx, void* qt, int64_t ldq, const float* scales, int R, int C, int dt, void* stream) {
  if (!x || !qt || !scales || R <= 0 || C <= 0 || (C & 7) || (ldx & 7) || (ldq & 15) || !aligned16(x) || !aligned16(qt)) return MH_ERR_ARG;
  if (ldq < (int64_t)(R + 127) / 128 * 128) return MH_ERR_ARG;
  const dim3 g2((C + 63) / 64, (R + 127) / 128);
  if (dt == MH_BF16)
    hipLaunchKernelGGL(quant_fp8_transposed_k<MH_BF16>, g2, dim3(256), 0, as_stream(stream), (const uint16_t*)x, ldx, scales, (uint8_t*)qt, ldq, R, C);
  else if (dt == MH_F16)
    hipLaunchKernelGGL(quant_fp8_transposed_k<MH_F16>, g2, dim3(256), 0, as_stream(stream), (const uint16_t*)x, ldx, scales, (uint8_t*)qt, ldq, R, C);
  else return MH_ERR_DTYPE;
  MH_LAUNCH_CHECK();
}
extern "C" int mh_quant_fp8_rows_e4(const void* w, int64_t ldw, void* q, float* scales, void* exps, int N, int K, int dt, void* stream) {
  if (!w || !q || !scales || !exps || N <= 0 || K <= 0 || (K & 127) || (ldw & 7) || !aligned16(w) || (((uintptr_t)q) & 7u) || K > 24576) return MH_ERR_ARG;
  const int64_t G = ((int64_t)(K / 128) * 64 + 4095) / 4096 * 4096;
  const dim3 grid(((N + 1) / 2 + 3) / 4), block(256);
  if (dt == MH_BF16)
    hipLaunchKernelGGL(quant_fp8_rows_e4_k<MH_BF16>, grid, block, 0, as_stream(stream), (const uint16_t*)w, ldw, (uint8_t*)q, scales, (uint8_t*)exps, G, N, K);
  else if (dt == MH_F16)
    hipLaunchKernelGGL(quant_fp8_rows_e4_k<MH_F16>, grid, block, 0, as_stream(stream), (const uint16_t*)w, ldw, (uint8_t*)q, scales, (uint8_t*)exps, G, N, K);
  else return MH_ERR_DTYPE;
  MH_LAUNCH_CHECK();
}

// w [R, C] -> qt [C, ldq >= round_up(R, 128)] = e4m3(w^T / (s[c] 2^-e[c, r / 128])), s [C], exponent image of the C output rows
extern "C" int mh_quant_fp8_rows_t_e4(const void* x, int64_t ldx, void* qt, int64_t ldq, float* scales, void* exps, unsigned* amax_ws, int R, int C,
                                      int dt, void* stream) {
  if (!x || !qt || !scales || !exps || !amax_ws || R <= 0 || C <= 0 || (C & 7) || (ldx & 7) || (ldq & 15) || !aligned16(x) || !aligned16(qt)) return MH_ERR_ARG;
  const int64_t Rp = (int64_t)(R + 127) / 128 * 128;
  if (ldq < Rp || Rp > 24576) return MH_ERR_ARG;
  if (dt != MH_BF16 && dt != MH_F16) return MH_ERR_DTYPE;
  const int64_t G = ((Rp / 128) * 64 + 4095) / 4096 * 4096;
  hipStream_t st = as_stream(stream);
  hipError_t e = hipMemsetAsync(amax_ws, 0, (size_t)C * sizeof(unsigned), st);
  if (e != hipSuccess) return (int)e;
  const int ncv = C / 8;
  int ry = (R + 63) / 64;
  if (ry > 512) ry = 512;
  const dim3 g1((ncv + 255) / 256, ry), g2((C + 63) / 64, (R + 127) / 128);
  if (dt == MH_BF16) {
    hipLaunchKernelGGL(col_absmax_k<MH_BF16>, g1, dim3(256), 0, st, (const uint16_t*)x, ldx, amax_ws, R, C);
    hipLaunchKernelGGL(amax_to_scale_k, dim3((C + 255) / 256), dim3(256), 0, st, (const unsigned*)amax_ws, scales, C);
    hipLaunchKernelGGL(quant_fp8_transposed_e4_k<MH_BF16>, g2, dim3(256), 0, st, (const uint16_t*)x, ldx, (const float*)scales, (uint8_t*)qt, ldq, (uint8_t*)exps, G, R, C);
  } else {
    hipLaunchKernelGGL(col_absmax_k<MH_F16>, g1, dim3(256), 0, st, (const uint16_t*)x, ldx, amax_ws, R, C);
    hipLaunchKernelGGL(amax_to_scale_k, dim3((C + 255) / 256), dim3(256), 0, st, (const unsigned*)amax_ws, scales, C);
    hipLaunchKernelGGL(quant_fp8_transposed_e4_k<MH_F16>, g2, dim3(256), 0, st, (const uint16_t*)x, ldx, (const float*)scales, (uint8_t*)qt, ldq, (uint8_t*)exps, G, R, C);
  }
  MH_LAUNCH_CHECK();
}

// x [R, C] -> (q [R, ldq] row-quantised, sr [R]) and (qt [C, ldqt >= round_up(R, 128)] column-quantised + transposed, sc [C]) with two
// reads of x (maxima, then both copies) instead of four; ws: R + C uints of scratch.
extern "C" int mh_quant_fp8_rows_and_t(const void* x, int64_t ldx, void* q, int64_t ldq, float* sr, void* qt, int64_t ldqt, float* sc,
                                       unsigned* ws, int R, int C, int dt, void* stream) {
  if (!x || !q || !sr || !qt || !sc || !ws || R <= 0 || C <= 0 || (C & 7) || (ldx & 7) || (ldq & 7) || (ldqt & 15) || !aligned16(x) || !aligned16(qt) ||
      (((uintptr_t)q) & 7u))
    return MH_ERR_ARG;
  if (ldqt < (int64_t)(R + 127) / 128 * 128 || ldq < C) return MH_ERR_ARG;
  if (dt != MH_BF16 && dt != MH_F16) return MH_ERR_DTYPE;
  hipStream_t st = as_stream(stream);
  hipError_t e = hipMemsetAsync(ws, 0, (size_t)(R + C) * sizeof(unsigned), st);
  if (e != hipSuccess) return (int)e;
  const dim3 grid((C + 127) / 128, (R + 127) / 128);
  unsigned *rmax = ws, *cmax = ws + R;
  if (dt == MH_BF16) {
    hipLaunchKernelGGL(absmax_rc_k<MH_BF16>, grid, dim3(256), 0, st, (const uint16_t*)x, ldx, rmax, cmax, R, C);
    hipLaunchKernelGGL(quant_fp8_both_k<MH_BF16>, grid, dim3(256), 0, st, (const uint16_t*)x, ldx, (const unsigned*)rmax, (const unsigned*)cmax, (uint8_t*)q, ldq, sr,
                       (uint8_t*)qt, ldqt, sc, R, C);
  } else {
    hipLaunchKernelGGL(absmax_rc_k<MH_F16>, grid, dim3(256), 0, st, (const uint16_t*)x, ldx, rmax, cmax, R, C);
    hipLaunchKernelGGL(quant_fp8_both_k<MH_F16>, grid, dim3(256), 0, st, (const uint16_t*)x, ldx, (const unsigned*)rmax, (const unsigned*)cmax, (uint8_t*)q, ldq, sr,
                       (uint8_t*)qt, ldqt, sc, R, C);
  }
  MH_LAUNCH_CHECK();
}
// The second half of mh_quant_fp8_rows_and_t for a tensor whose maxima are already known: ws = [R row maxima | C column maxima] as bit patterns of
// non-negative floats (what absmax_rc_k leaves there; mh_gemm_fp8_swiglu_bwd_amax produces them in the GEMM's store phase) - ONE read of x.
extern "C" int mh_quant_fp8_rows_and_t_pre(const void* x, int64_t ldx, void* q, int64_t ldq, float* sr, void* qt, int64_t ldqt, float* sc,
                                           const unsigned* ws, int R, int C, int dt, void* stream) {
  if (!x || !q || !sr || !qt || !sc || !ws || R <= 0 || C <= 0 || (C & 7) || (ldx & 7) || (ldq & 7) || (ldqt & 15) || !aligned16(x) || !aligned16(qt) ||
      (((uintptr_t)q) & 7u))
    return MH_ERR_ARG;
  if (ldqt < (int64_t)(R + 127) / 128 * 128 || ldq < C) return MH_ERR_ARG;
  if (dt != MH_BF16 && dt != MH_F16) return MH_ERR_DTYPE;
  const dim3 grid((C + 127) / 128, (R + 127) / 128);
  const unsigned *rmax = ws, *cmax = ws + R;
  if (dt == MH_BF16)
    hipLaunchKernelGGL(quant_fp8_both_k<MH_BF16>, grid, dim3(256), 0, as_stream(stream), (const uint16_t*)x, ldx, rmax, cmax, (uint8_t*)q, ldq, sr, (uint8_t*)qt, ldqt, sc, R, C);
  else
    hipLaunchKernelGGL(quant_fp8_both_k<MH_F16>, grid, dim3(256), 0, as_stream(stream), (const uint16_t*)x, ldx, rmax, cmax, (uint8_t*)q, ldq, sr, (uint8_t*)qt, ldqt, sc, R, C);
  MH_LAUNCH_CHECK();
}
// row and column maxima of |x| into zeroed rmax [R] / cmax [C] (gemm.hip: the 8-wave fallback of mh_gemm_fp8_swiglu_bwd_amax)
namespace mhgemm {
int launch_absmax_rc(const void* x, int64_t ldx, unsigned* rmax, unsigned* cmax, int R, int C, int dt, hipStream_t stream) {
  if (!x || !rmax || !cmax || (C & 7) || (ldx & 7) || !aligned16(x)) return MH_ERR_ARG;
  const dim3 grid((C + 127) / 128, (R + 127) / 128);
  if (dt == MH_BF16) hipLaunchKernelGGL(absmax_rc_k<MH_BF16>, grid, dim3(256), 0, stream, (const uint16_t*)x, ldx, rmax, cmax, R, C);
  else if (dt == MH_F16) hipLaunchKernelGGL(absmax_rc_k<MH_F16>, grid, dim3(256), 0, stream, (const uint16_t*)x, ldx, rmax, cmax, R, C);
  else return MH_ERR_DTYPE;
  MH_LAUNCH_CHECK();
}
}  // namespace mhgemm
extern "C" int mh_max_to_vec(const float* s, int n, float* out, int m, void* stream) {
  if (!s || !out || n <= 0 || m <= 0) return MH_ERR_ARG;
  hipLaunchKernelGGL(max_to_vec_k, dim3(1), dim3(1024), 0, as_stream(stream), s, n, out, m);
  MH_LAUNCH_CHECK();
}

// x [R, C] (16-bit, row stride ldx elements) -> qt [C, ldq bytes] with ldq >= round_up(R, 128) (columns R.. of qt up to the
// next multiple of 128 are zero-filled), scales [C]; amax_ws: C uints of scratch (zeroed here).
extern "C" int mh_quant_fp8_rows_t(const void* x, int64_t ldx, void* qt, int64_t ldq, float* scales, unsigned* amax_ws, int R, int C, int dt,
                                   void* stream) {
  if (!x || !qt || !scales || !amax_ws || R <= 0 || C <= 0 || (C & 7) || (ldx & 7) || (ldq & 15) || !aligned16(x) || !aligned16(qt)) return MH_ERR_ARG;
  if (ldq < (int64_t)(R + 127) / 128 * 128) return MH_ERR_ARG;
  if (dt != MH_BF16 && dt != MH_F16) return MH_ERR_DTYPE;
  hipStream_t st = as_stream(stream);
  hipError_t e = hipMemsetAsync(amax_ws, 0, (size_t)C * sizeof(unsigned), st);
  if (e != hipSuccess) return (int)e;
  const int ncv = C / 8;
  int ry = (R + 63) / 64;
  if (ry > 512) ry = 512;
  const dim3 g1((ncv + 255) / 256, ry), g2((C + 63) / 64, (R + 127) / 128);
  if (dt == MH_BF16) {
    hipLaunchKernelGGL(col_absmax_k<MH_BF16>, g1, dim3(256), 0, st, (const uint16_t*)x, ldx, amax_ws, R, C);
    hipLaunchKernelGGL(amax_to_scale_k, dim3((C + 255) / 256), dim3(256), 0, st, (const unsigned*)amax_ws, scales, C);
    hipLaunchKernelGGL(quant_fp8_transposed_k<MH_BF16>, g2, dim3(256), 0, st, (const uint16_t*)x, ldx, (const float*)scales, (uint8_t*)qt, ldq, R, C);
  } else {
    hipLaunchKernelGGL(col_absmax_k<MH_F16>, g1, dim3(256), 0, st, (const uint16_t*)x, ldx, amax_ws, R, C);
    hipLaunchKernelGGL(amax_to_scale_k, dim3((C + 255) / 256), dim3(256), 0, st, (const unsigned*)amax_ws, scales, C);
    hipLaunchKernelGGL(quant_fp8_transposed_k<MH_F16>, g2, dim3(256), 0, st, (const uint16_t*)x, ldx, (const float*)scales, (uint8_t*)qt, ldq, R, C);
  }
  MH_LAUNCH_CHECK();
}
