/* Development-only A/B arms (NOT part of the product library or its ABI): the first-generation attention kernels that
 * need a V re-layout pass / an operand workspace, and the alternative GEMM tilings reached through
 * mh_gemm_force_kernel(32 | 4..12 | 88).  Built by `python -m merlin_amd.csrc.build --dev` into
 * tools/dev_arms/libmerlin_hip_dev.so (product sources compiled with -DMH_DEV_ARMS + the files in this directory). */
#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* vt[b, h, d, perm(s)] = qkv[(b*S + s), which=2, h, d]; S_pad = round_up(S, 64); perm swaps
 * bits 2 and 3 of s (the MFMA k-slot order the attention kernels use). Tail is zero filled. */
int mh_attn_prep_v(const void* v, int64_t ldv, void* vt, int B, int S, int H, int D, int dt, void* stream);
/* Flash attention forward.  q/k are [B*S, H, D] views with row stride ldq/ldk (elements);
 * vt from mh_attn_prep_v; o is [B*S, H*D] (ldo); lse fp32 [B, H, S_pad].  seqlens int32[B] or null
 * (= S): keys >= seqlens[b] are excluded and query rows >= seqlens[b] are written as zeros,
 * i.e. flash_attn_varlen + pad_input semantics (llama_flash_attn_monkey_patch.py:87-102).
 * causal=1: Llama (D=128), causal=0: CLIP (D=64).  scale = 1/sqrt(D). */
int mh_attn_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, void* o, int64_t ldo,
                float* lse, const int32_t* seqlens, int B, int S, int H, int D, int causal, int dt, void* stream);
/* Backward: `delta` is a ZERO-INITIALISED fp32 scratch of 2*B*H*S_pad floats (rowsum(dO*O), then lse*log2e).
 * dq/dk/dv are [B*S, H, D] views with their own row strides.  v is the ROW-MAJOR v (not vt).
 * ws: 16-bit workspace of mh_attn_bwd_ws_elems() elements (holds Q^T, dO^T, K^T re-layouts). */
int64_t mh_attn_bwd_ws_elems(int B, int S, int H, int D);
/* A/B switch for benchmarks: 1 (default) = separate dV and dK launches (lean kernels), 0 = one fused launch */
void mh_attn_bwd_split(int split);
int mh_attn_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                const void* o, int64_t ldo, const void* dout, int64_t lddo, const float* lse, float* delta,
                void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, void* ws,
                const int32_t* seqlens, int B, int S, int H, int D, int causal, int dt, void* stream);

#ifdef __cplusplus
}
#endif
