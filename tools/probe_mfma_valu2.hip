// Follow-up of probe_mfma_valu.hip: the exact slot pattern of the pipelined forward (csrc/attn_fwd3.hip), built up step by step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// V: 1 = fma,exp,fma per MFMA; 2 = the pair pattern; 3 = + LDS transpose reads and lgkm waits; 4 = 3 with accumulator reuse distance 2;
//    5 = 4 without MFMAs; 6 = 4 with independent (non-chained) VALU registers
template <int V>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, long long* cyc) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  for (int i = threadIdx.x; i < 16384; i += 256) ((float*)lds)[i] = 0.001f * i;
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  u32x4 a = {0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, b = a;
  float st[32];
  for (int i = 0; i < 32; ++i) st[i] = 0.001f * (threadIdx.x + i);
  const float s = 0.9999f, t = -1.0f;
  f32x2 ps = {0.f, 0.f};
  unsigned pk[16];
  for (int i = 0; i < 16; ++i) pk[i] = 0;
  const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + (threadIdx.x & 63) * 8;
  u32x2 w[8];
  for (int i = 0; i < 8; ++i) w[i] = u32x2{0x3c003c00u, 0x3c003c00u};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    float t1 = 0.f, e0 = 0.f;
    float fa[4] = {0.f, 0.f, 0.f, 0.f}, fb[4] = {0.f, 0.f, 0.f, 0.f}, ea[4] = {0.f, 0.f, 0.f, 0.f}, eb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 64; ++m) {
      constexpr int dummy = 0;
      const int ai = (V >= 4) ? (m & 1) : (m & 3);
      if (V >= 3 && (m & 1) == 0) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
      if (V != 5 && V != 8) {
        if (V >= 3) {
          const u32x4 af = {w[2 * ((m / 2) & 3)][0], w[2 * ((m / 2) & 3)][1], w[2 * ((m / 2) & 3) + 1][0], w[2 * ((m / 2) & 3) + 1][1]};
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[ai]) : "v"(af), "v"(b));
        } else {
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[ai]) : "v"(a), "v"(b));
        }
      }
      if (V >= 3 && (m & 1) == 1) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "=v"(w[2 * ((m / 2) & 3)]) : "v"(addr));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(w[2 * ((m / 2) & 3) + 1]) : "v"(addr));
      }
      if (V == 7 || V == 8) {
        // even slot: two fma of pair p+1, cvt_pk of pair p-1; odd slot: two exp of pair p, pk_add of pair p-1
        if ((m & 1) == 0) {
          asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(fa[((m / 2) + 1) & 3]) : "v"(st[m & 31]), "v"(s), "v"(t));
          asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(fb[((m / 2) + 1) & 3]) : "v"(st[(m + 1) & 31]), "v"(s), "v"(t));
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[(m / 2) & 15]) : "v"(ea[((m / 2) + 3) & 3]), "v"(eb[((m / 2) + 3) & 3]));
        } else {
          asm volatile("v_exp_f32 %0, %1" : "=v"(ea[(m / 2) & 3]) : "v"(fa[(m / 2) & 3]));
          asm volatile("v_exp_f32 %0, %1" : "=v"(eb[(m / 2) & 3]) : "v"(fb[(m / 2) & 3]));
          f32x2 ee = {ea[((m / 2) + 3) & 3], eb[((m / 2) + 3) & 3]};
          asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(ps) : "v"(ee));
        }
      } else if (V == 1) {
        float x0, x1, y;
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x0) : "v"(st[m & 31]), "v"(s), "v"(t));
        asm volatile("v_exp_f32 %0, %1" : "=v"(y) : "v"(x0));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x1) : "v"(st[(m + 1) & 31]), "v"(s), "v"(t));
        st[m & 31] = y + x1;
      } else if ((m & 1) == 0) {
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t1) : "v"(st[m & 31]), "v"(s), "v"(t));
        float x0;
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x0) : "v"(st[(m + 1) & 31]), "v"(s), "v"(t));
        asm volatile("v_exp_f32 %0, %1" : "=v"(e0) : "v"(x0));
      } else {
        float e1;
        asm volatile("v_exp_f32 %0, %1" : "=v"(e1) : "v"(t1));
        f32x2 ee = {e0, e1};
        if (V == 6) {
          asm volatile("v_pk_add_f32 %0, %1, %1" : "=v"(ee) : "v"(ee));
        } else {
          asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(ps) : "v"(ee));
        }
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[(m / 2) & 15]) : "v"(e0), "v"(e1));
      }
    }
  }
  long long t1c = __builtin_readcyclecounter();
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]));
  float r = ps[0] + ps[1];
  for (int i = 0; i < 4; ++i) r += acc[i][0];
  for (int i = 0; i < 32; ++i) r += st[i];
  for (int i = 0; i < 16; ++i) r += (float)pk[i];
  for (int i = 0; i < 8; ++i) r += (float)w[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1c - t0;
}

template <int V>
void run(const char* name, float* out, long long* cyc) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<V><<<256, 256>>>(out, 10, cyc);
  hipEventRecord(e0);
  k<V><<<256, 256>>>(out, iters, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-72s %8.3f ms  %8.1f cycles per 64 slots\n", name, ms, (double)c / iters);
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  run<1>("1: MFMA + fma, exp, fma", out, cyc);
  run<2>("2: pair pattern (fma fma exp | exp pk_add cvt_pk)", out, cyc);
  run<3>("3: 2 + two ds_read_b64_tr per pair + lgkmcnt wait, A operand from the window", out, cyc);
  run<4>("4: 3 with accumulator reuse distance 2", out, cyc);
  run<5>("5: 4 without the MFMAs", out, cyc);
  run<6>("6: 4 without the pk_add chain", out, cyc);
  run<7>("7: 4 with the VALU stream software-pipelined (no back-to-back dependence)", out, cyc);
  run<8>("8: 7 without the MFMAs", out, cyc);
  return 0;
}
