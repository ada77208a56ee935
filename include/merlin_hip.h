/*
 * merlin_hip.h - C ABI of libmerlin_hip.so, the MI355X (gfx950) kernels behind the Merlin /
 * MMGPT hot path (MMGPTLlamaForCausalLM.forward + backward).
 *
 * The reference (Ahnsun/merlin) has NO native/FFI boundary: its hot path is Python calling
 * torch / transformers / flash-attn.  This header therefore does not replace
 * an existing FFI; each entry point names the reference call site whose GPU work it takes
 * over (file:line under /root/reference).  INTEGRATION.md shows the Python binding a
 * reference maintainer would add.
 *
 * Conventions
 *  - plain pointers + sizes only; all pointers are DEVICE pointers unless stated otherwise.
 *  - the caller allocates everything (outputs, workspaces); the library never allocates,
 *    frees or synchronises.  Work is enqueued on `stream` (a hipStream_t passed as void*).
 *  - every function returns 0 on success, a negative MH_ERR_* on bad arguments, or a positive
 *    hipError_t from the launch.
 *  - `dt` is the 16-bit storage/MFMA type of activations and weights: MH_BF16 or MH_F16.
 *    Accumulation, softmax, norms statistics and losses are always fp32.
 *  - matrices are row-major with explicit leading dimensions in ELEMENTS.
 */
#ifndef MERLIN_HIP_H
#define MERLIN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MH_BF16 0
#define MH_F16 1
#define MH_F32 2

#define MH_OK 0
#define MH_ERR_ARG (-1)    /* bad shape / alignment / null pointer */
#define MH_ERR_DTYPE (-2)  /* unsupported dtype */
#define MH_ERR_ARCH (-3)   /* device is not gfx950 */
#define MH_ERR_SHAPE (-4)  /* shape not supported by this kernel (e.g. head_dim) */

/* epilogue flags for mh_gemm_nt */
#define MH_EPI_BIAS 1        /* += bias[n]                                   */
#define MH_EPI_QUICK_GELU 2  /* x * sigmoid(1.702 x)  (CLIP fc1, modeling_clip quick_gelu) */
#define MH_EPI_RESIDUAL 4    /* += resid[m, n]                               */
#define MH_EPI_ACCUM 8       /* C = C_old + result (wgrad accumulation)      */
#define MH_EPI_OUT_F32 16    /* C is fp32 instead of `dt`                    */

int mh_version(void);
/* 1 if device `dev` is gfx950, else 0 (host query, no stream). */
int mh_arch_ok(int dev);
const char* mh_strerror(int code);

/* ---- deterministic weight generator (merlin_amd/weights.py, bit-identical) ------------- */
/* out[i] = round_bf16(offset + sigma * irwin_hall(key, start + i)), stored as `dt` (incl. MH_F32). */
int mh_fill_normal(void* out, int64_t n, uint64_t key, int64_t start, float sigma, float offset, int dt, void* stream);

/* ---- GEMM:  C[M,N] = A[M,K] * B[N,K]^T (+epilogue)  ------------------------------------
 * Replaces every nn.Linear / conv-as-GEMM on the path: q/k/v/o_proj, gate/up/down_proj,
 * lm_head (llama_flash_attn_monkey_patch.py:35-49,103; llama_mmgpt.py:87), CLIP q/k/v/out,
 * fc1/fc2, patch embedding (clip_encoder.py:79) and the projector (mlp_projector.py:22).
 * A, B are `dt`; K % 64 == 0; lda/ldb % 8 == 0; 16-byte aligned bases.  bf16/f16 MFMA, fp32
 * accumulate.  bias is `dt`[N]; resid is `dt`[M, ldr]. */
int mh_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
               const void* bias, const void* resid, int64_t ldr,
               int M, int N, int K, int dt, int epilogue, void* stream);

/* General form: each operand is K-contiguous (x_kstrided = 0: A[M,K] / B[N,K]) or K-strided (1: A[K,M] / B[K,N],
 * i.e. the non-contracted index is the contiguous one).  C[M,N] = op(A) * op(B)^T as above.  Backward GEMMs need
 * no transposed copies: dgrad dX[T,Kin] = dY[T,Nout] * W[Nout,Kin] is (A=dY, B=W K-strided); wgrad dW[Nout,Kin] =
 * dY^T * X is (A=dY K-strided, B=X K-strided, K = T).  K-strided operands need M (resp. N) % 8 == 0; K % 64 == 0
 * unless BOTH operands are K-strided (then any K: the tail rows are read as zeros). */
int mh_gemm(const void* A, int64_t lda, int a_kstrided, const void* B, int64_t ldb, int b_kstrided, void* C, int64_t ldc,
            const void* bias, const void* resid, int64_t ldr, int M, int N, int K, int dt, int epilogue, void* stream);

/* Split-K form for weight gradients whose output is small next to the contraction (CLIP tower: dW[<=4096, <=4096]
 * over K = 27 696 tokens gives only 16-64 output tiles for 256 CUs): `splits` blocks share an output tile, each
 * reduces its own K range into an fp32 partial tile in `ws` (splits * M * N floats), then one pass adds the
 * partials (fixed order: deterministic) into C (`dt` or fp32, optionally += C).  mh_gemm_splitk_max returns the
 * split count the library would use for a shape (1 = not worth splitting).  With both operands K-strided, K may be
 * any positive value: rows k >= K of the last K-tile are read as zeros. */
/* C = A B^T (16-bit, K-contiguous operands) with RoPE fused into the epilogue: columns [0, rope_cols) of C are heads of D
 * channels rotated (rotate-half) at position row % S with the mh_rope_table layout - the fused q|k|v projection of
 * llama_flash_attn_monkey_patch.py:35-59 in one launch, bit-identical to mh_gemm_nt followed by mh_rope_qk.
 * D in {64, 128}; rope_cols % D == 0; N % 8 == 0; 16-byte aligned C rows. */
int mh_gemm_nt_rope(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K, int dt,
                    const float* cos_sin, int S, int D, int rope_cols, void* stream);
/* SwiGLU (HF LlamaMLP, modeling_llama.py:174-176) fused into the neighbouring GEMM's epilogue; both forms are bit-identical
 * to the unfused sequence (GEMM, then mh_swiglu_fwd / mh_swiglu_bwd on the stored 16-bit tensor).
 * fwd: gu[M, 2ff] = x Wgu^T (Wgu = [gate; up] rows) AND act[M, ff] = silu(gate) * up in one launch.
 * bwd: dgu[M, 2ff] = swiglu'(gu) applied to dact = dy Wd, Wd = down_proj.weight [K = d_model, ff]; dact is never stored. */
int mh_gemm_swiglu_fwd(const void* x, int64_t ldx, const void* Wgu, int64_t ldw, void* gu, int64_t ldgu, void* act, int64_t ldact,
                       int M, int ff, int K, int dt, void* stream);
int mh_gemm_swiglu_bwd(const void* dy, int64_t lddy, const void* Wd, int64_t ldw, const void* gu, int64_t ldgu, void* dgu,
                       int64_t lddgu, int M, int ff, int K, int dt, void* stream);
/* fp8 GEMM (forward / inference form of BASELINE cfg 5's "fp8 MFMA weight path"): both operands OCP e4m3 bytes with ONE
 * fp32 scale per row (activation row = token, weight row = output channel), products on the gfx950 scaled-fp8 MFMA
 * (v_mfma_scale_f32_16x16x128_f8f6f4, hardware block scales 1.0), fp32 accumulate, C[m,n] = sa[m] sb[n] sum_k qa qb (+ the
 * usual epilogue) in `dt_out`.  mh_quant_fp8_rows produces (q, scales) from a 16-bit matrix: scale = max|row| / 448.
 * K % 128 == 0; lda, ldb in bytes, % 16 == 0. */
int mh_quant_fp8_rows(const void* x, int64_t ldx, void* q, int64_t ldq, float* scales, int R, int K, int dt, void* stream);
int mh_gemm_fp8(const void* A8, int64_t lda, const float* sa, const void* B8, int64_t ldb, const float* sb, const void* b_exp, void* C,
                int64_t ldc, const void* bias, const void* resid, int64_t ldr, int M, int N, int K, int dt_out, int epilogue, void* stream);
/* fp8 forms of mh_gemm_nt_rope and mh_gemm_swiglu_fwd (the same fused store phases behind the fp8 product). */
int mh_gemm_fp8_rope(const void* A8, int64_t lda, const float* sa, const void* B8, int64_t ldb, const float* sb, const void* b_exp, void* C,
                     int64_t ldc, int M, int N, int K, int dt_out, const float* cos_sin, int S, int D, int rope_cols, void* stream);
int mh_gemm_fp8_swiglu_fwd(const void* A8, int64_t lda, const float* sa, const void* B8, int64_t ldb, const float* sb, const void* b_exp,
                           void* gu, int64_t ldgu, void* act, int64_t ldact, int M, int ff, int K, int dt_out, void* stream);
/* b_exp (nullable, all four fp8 GEMM entry points): per-128-block scales of the B operand (the weights) - BASELINE cfg 5's
 * "per-128-block scales" - as 4-bit exponents e, block scale = sb[n] * 2^-e, applied by the MFMA's own E8M0 block-scale operand
 * (v_mfma_scale_f32_16x16x128_f8f6f4) at no cost in the K loop.  Layout produced by mh_quant_fp8_rows_e4 / mh_quant_fp8_rows_t_e4:
 * a 16-byte header (int32 flag "some exponent is non-zero", raised by the quantiser - the GEMM takes its constant-scale loop while it is
 * 0, e.g. for i.i.d. initialised weights - then padding; ZERO the header before quantising) followed by [ceil(N / 256) * 2 groups of
 * 128 rows][G bytes], G = round_up(K / 128 * 64, 4096), a group = [K / 128 blocks][64 B], two rows per byte (low nibble = even row).
 * K <= 24 576. */
int mh_quant_fp8_rows_e4(const void* w, int64_t ldw, void* q, float* scales, void* exps, int N, int K, int dt, void* stream);
int mh_quant_fp8_rows_t_e4(const void* w, int64_t ldw, void* qt, int64_t ldq, float* scales, void* exps, unsigned* amax_ws, int R, int C, int dt,
                           void* stream);
/* ---- fp8 TRAINING step (BASELINE cfg 5: fp8 MFMA weight path).  Every GEMM is an NT product of two row-quantised operands:
 *   forward  y  = x W^T        rowquant(x)    [T, K]   x rowquant(W)     [N, K]
 *   dgrad    dx = dy W         rowquant(dy)   [T, N]   x rowquant(W^T)   [K, N]
 *   wgrad    dW = dy^T x       rowquant(dy^T) [N, Tp]  x rowquant(x^T)   [K, Tp]   (Tp = tokens rounded up to 128, zero fill)
 * mh_quant_fp8_rows_t produces the transposed operands: qt[c, r] = e4m3(x[r, c] / s[c]), s[c] = max_r |x[r, c]| / 448, from a
 * 16-bit x [R, C] (row stride ldx elements); qt rows are ldq bytes (>= round_up(R, 128), bytes beyond R zero); amax_ws = C
 * uints of scratch.  mh_gemm_fp8 accepts MH_EPI_ACCUM (gradient accumulation into the 16-bit gradient arena).
 * mh_gemm_fp8_swiglu_bwd: dgu = swiglu_bwd(gu, dy Wd) with WdT8 = rowquant(down_proj.weight^T) [ff, d_model]. */
int mh_quant_fp8_rows_t(const void* x, int64_t ldx, void* qt, int64_t ldq, float* scales, unsigned* amax_ws, int R, int C, int dt, void* stream);
/* Single-pass form used by the training step for activations / gradients: the transposed copy is scaled by a [C] vector the
 * caller provides - in practice ONE tensor-wide scale, the maximum of the tensor's row scales (mh_max_to_vec fills out[0..m)
 * with max_i s[i]), which mh_quant_fp8_rows already produced: no column-maximum pass over the data. */
int mh_quant_fp8_t_scaled(const void* x, int64_t ldx, void* qt, int64_t ldq, const float* scales, int R, int C, int dt, void* stream);
int mh_max_to_vec(const float* s, int n, float* out, int m, void* stream);
/* Both operand forms of a gradient tensor from TWO reads of it (row + column maxima by order-independent atomicMax in one pass over
 * 128 x 128 tiles, then the row-quantised copy q [R, ldq] / sr [R] and the transposed column-quantised copy qt [C, ldqt] / sc [C]
 * from one more): dgrad consumes (q, sr), wgrad (qt, sc).  ws: R + C uints of scratch (zeroed here). */
int mh_quant_fp8_rows_and_t(const void* x, int64_t ldx, void* q, int64_t ldq, float* sr, void* qt, int64_t ldqt, float* sc, unsigned* ws,
                            int R, int C, int dt, void* stream);
int mh_gemm_fp8_swiglu_bwd(const void* dy8, int64_t lddy, const float* sdy, const void* WdT8, int64_t ldw, const float* swt, const void* wt_exp,
                           const void* gu, int64_t ldgu, void* dgu, int64_t lddgu, int M, int ff, int K, int dt_out, void* stream);
/* mh_gemm_fp8_swiglu_bwd that also leaves, in amax_ws [M + 2 ff] (zeroed here), the bit patterns of max |dgu| per row and then per column of the tensor as
 * stored: taken by the GEMM's store phase itself (order-independent atomicMax) where the 4-wave fp8 kernel runs, by one read of dgu otherwise.
 * mh_quant_fp8_rows_and_t_pre = mh_quant_fp8_rows_and_t without its maxima pass, fed from such a buffer: together they save one read of the largest
 * gradient tensor of a decoder layer ([tokens, 2 ff]) per layer and step; the bytes written are identical to the two-pass form's. */
int mh_gemm_fp8_swiglu_bwd_amax(const void* dy8, int64_t lddy, const float* sdy, const void* WdT8, int64_t ldw, const float* swt, const void* wt_exp,
                                const void* gu, int64_t ldgu, void* dgu, int64_t lddgu, unsigned* amax_ws, int M, int ff, int K, int dt_out, void* stream);
int mh_quant_fp8_rows_and_t_pre(const void* x, int64_t ldx, void* q, int64_t ldq, float* sr, void* qt, int64_t ldqt, float* sc, const unsigned* ws,
                                int R, int C, int dt, void* stream);
int mh_gemm_splitk_max(int M, int N, int K);
int mh_gemm_splitk(const void* A, int64_t lda, int a_kstrided, const void* B, int64_t ldb, int b_kstrided, void* C,
                   int64_t ldc, int M, int N, int K, int dt, int accumulate, int out_f32, int splits, float* ws,
                   void* stream);

/* CLIP MLP (HF CLIPMLP fc1 -> quick_gelu -> fc2, called from clip_encoder.py:79): quick-GELU in the GEMM's store phase.  Forward: f1 [M, N] =
 * x [M, K] W1[N, K]^T + b1 and a = f1 * sigmoid(1.702 f1) (both stored: the backward needs f1).  Backward: df1 [M, N] = quick_gelu'(f1) * (dy [M, K] w2[K, N])
 * with w2 = fc2.weight read K-strided and dy w2 never stored.  Element maths on the rounded 16-bit tile: bit-identical to mh_gemm + mh_quick_gelu_fwd / _bwd.
 * N, ldf, lda_out, lddf multiples of 8, 16-byte aligned outputs. */
int mh_gemm_gelu_fwd(const void* x, int64_t ldx, const void* w1, int64_t ldw, const void* bias, void* f1, int64_t ldf, void* a, int64_t lda_out,
                     int M, int N, int K, int dt, void* stream);
int mh_gemm_gelu_bwd(const void* dy, int64_t lddy, const void* w2, int64_t ldw, const void* f1, int64_t ldf, void* df1, int64_t lddf,
                     int M, int N, int K, int dt, void* stream);

/* Several weight gradients over the SAME token count in one launch: out_p[M_p, N_p] (+)= dy_p[T, M_p]^T x_p[T, N_p] for p < n <= 8, both
 * operands as they lie in memory (K-strided), any T (rows >= T read as zeros), 16-bit outputs.  One block per 256 x 256 output tile of
 * any of the problems: the four Linears of a CLIP encoder layer (48 + 16 + 64 + 64 tiles; autograd of HF CLIPEncoderLayer,
 * clip_encoder.py:74-82) fill the chip together for the whole contraction where each of them alone needs split-K with fp32 partials.
 * Returns MH_ERR_SHAPE when a problem does not meet the 4-wave kernel's conditions (M, N, ld % 8, 16-byte aligned bases, operand
 * spans < 4 GiB): nothing is launched then and the caller issues mh_gemm_splitk per problem. */
typedef struct MhWgradProblem {
  const void* dy; int64_t lddy;   /* [T, M] */
  const void* x;  int64_t ldx;    /* [T, N] */
  void* out;      int64_t ldo;    /* [M, N], 16-bit */
  int M, N, accumulate, reserved;
} MhWgradProblem;
int mh_wgrad_grouped(const MhWgradProblem* problems, int n, int T, int dt, void* stream);

/* Split-K with mh_gemm's full epilogue (bias, quick-GELU, residual, accumulate, fp32 store - applied by the fixed-order reduce pass
 * exactly as the one-pass store phase applies them): for products with few output tiles and a long contraction, e.g. the o / down
 * projections and the dgrads of a 613-token sequence (48 tiles for 256 CUs).  ws: splits * M * N floats; N % 4 == 0, ldc % 4 == 0;
 * splits from mh_gemm_splitk_max (1 = the plain mh_gemm call).  Replaces the same nn.Linear calls as mh_gemm. */
int mh_gemm_splitk_epi(const void* A, int64_t lda, int a_kstrided, const void* B, int64_t ldb, int b_kstrided, void* C, int64_t ldc,
                       const void* bias, const void* resid, int64_t ldr, int M, int N, int K, int dt, int epilogue, int splits, float* ws,
                       void* stream);

/* Kernel selection override for tests / A-B benchmarks: 0 = auto (256x256 tiles when they fill the chip, else 128x128; among the
 * 256x256 kernels the 4-wave form where mh_gemm_w4_policy says so), 128 = the 128x128 kernel, 256 = the 8-wave 256x256 kernel,
 * 4 = the 4-wave 256x256 kernel (128x128 outputs per wave, csrc/gemm_w4.hip) wherever it can run. */
void mh_gemm_force_kernel(int which);
/* Operand layouts the auto selection gives to the 4-wave kernel when K >= 4096 and the epilogue is a plain (or accumulating)
 * 16-bit store: bit 0 = TN (weight gradients: both operands K-strided), bit 1 = NN (dgrad), bit 2 = NT (forward; also with a
 * residual), bit 3 = the fp8 training step's exponent-free NT products (gemm_w4_f8).  Default 11 (measured: profiles/r03_gemm_w4_ab.txt). */
void mh_gemm_w4_policy(int mask);
/* 128-row block tiles of the 4-wave kernel (a wave owns 64 x 128 outputs) for NT products with few rows - a 613-token prefill
 * (BASELINE configs[0]/[1]; every `generate` prompt, llama_mmgpt.py:114-134) is 2.4 tiles of 256 rows: 0 = never, 1 (default) = where
 * they take fewer rounds of the 256 CUs than 256-row tiles, 2 = wherever the form exists (tests / A-B). */
void mh_gemm_w4_half(int mode);
/* 256x256-tile kernels: 1 (default) = persistent launch, one block per CU looping over the output tiles with the next tile's
 * first K-tile fetched under the epilogue; 0 = one block per tile (A-B benchmarks). */
void mh_gemm_persistent(int on);
/* Tile raster of the 256x256 / 128x128 kernels: output tiles are visited in groups of `gm` tile-rows x all tile-columns, and each XCD
 * gets a contiguous run of the sequence (32 tiles per round = gm x 32/gm): default 4 (A-B benchmarks: 2..16). */
void mh_gemm_raster_group(int gm);

/* out[C, R_pad] = in[R, C]^T for 16-bit elements (operand re-layout for dgrad / wgrad GEMMs);
 * columns [R, R_pad) of out are zero filled so the transposed operand's K is a multiple of 64. */
int mh_transpose16(const void* in, int64_t ldi, void* out, int64_t ldo, int R, int C, int R_pad, void* stream);

/* ---- norms -------------------------------------------------------------------------- */
/* LlamaRMSNorm (transformers modeling_llama.py LlamaRMSNorm; called 65x per forward). */
int mh_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd_or_null, int rows, int d, float eps, int dt, void* stream);
/* RMSNorm forward + the row-quantised e4m3 copy (q [rows, d] bytes, scales [rows]) of its output in the same launch: bit-identical
 * to mh_rmsnorm_fwd followed by mh_quant_fp8_rows (fp8 training / inference paths). */
int mh_rmsnorm_fwd_q8(const void* x, const void* w, void* y, void* q, float* scales, int rows, int d, float eps, int dt, void* stream);
/* dx (and fp32 partial dw[nblk, d], nblk = mh_norm_bwd_partials(rows)) */
int mh_rmsnorm_bwd(const void* x, const void* w, const void* dy, void* dx, float* dw_partial, int rows, int d, float eps, int dt, int accumulate_dx, void* stream);
/* nn.LayerNorm (CLIP pre_layrnorm / layer_norm1 / layer_norm2, eps 1e-5). */
int mh_layernorm_fwd(const void* x, const void* w, const void* b, void* y, int rows, int d, float eps, int dt, void* stream);
/* fp32 RESIDUAL STREAM (opt-in, engine.fp32_residual): x is fp32 [rows, d], updated in place by the accumulating fp32 epilogue of the
 * projections that feed it (mh_gemm with MH_EPI_OUT_F32 | MH_EPI_ACCUM [| MH_EPI_BIAS]).  y = RMSNorm(x; w) (b == NULL) or
 * LayerNorm(x; w, b) in `dt` for the next GEMM; x16 (nullable) receives the 16-bit copy of x the backward keeps as the layer input.
 * Removes every 16-bit rounding of HF's residual adds (LlamaDecoderLayer: `hidden_states = residual + hidden_states`) from the
 * forward: BASELINE's "logits within 1e-3 rel fp16" at depth is limited by exactly those (profiles/r01_full_depth_rounding_attribution.txt). */
int mh_norm_fwd_f32in(const float* x, const void* w, const void* b, void* y, void* x16, int rows, int d, float eps, int dt, void* stream);
/* fp32 tensors at the START of the fp32 residual streams (engine.fp32_residual): the patch projection (mh_gemm with MH_EPI_OUT_F32), class / position
 * embeddings added in fp32 (mh_vit_assemble_f32), pre_layrnorm from fp32 to fp32 (mh_layernorm_f32_to_f32; x16 = 16-bit copy of its input for the
 * backward), the projector's fp32 output spliced with widened embedding rows (mh_embed_splice_fwd_f32) - HF CLIPVisionEmbeddings / pre_layrnorm
 * (clip_encoder.py:79) and base_mmgpt.py:99-160 without a 16-bit rounding before the first residual add. */
int mh_vit_assemble_f32(const float* patch, const void* cls, const void* pos, float* x, int N, int G2, int d, int dt, void* stream);
int mh_layernorm_f32_to_f32(const float* x, const void* w, const void* b, float* y, void* x16, int rows, int d, float eps, int dt, void* stream);
int mh_embed_splice_fwd_f32(const int64_t* ids, const int32_t* src, const void* embed, const float* feats, float* out, int T, int d, int dt, void* stream);
int mh_layernorm_bwd(const void* x, const void* w, const void* dy, void* dx, float* dw_partial, float* db_partial, int rows, int d, float eps, int dt, int accumulate_dx, void* stream);
int mh_norm_bwd_partials(int rows);
/* out[d] (dt) (+)= sum_r partial[r, d]  (finishes dw/db; also bias grads from mh_colsum) */
int mh_reduce_partials(const float* partial, int nblk, int d, void* out, int dt, int accumulate, void* stream);
/* partial[nblk, d] = per-row-block column sums of x[rows, d] (bias gradients). nblk = mh_norm_bwd_partials(rows) */
int mh_colsum_partial(const void* x, int64_t ldx, float* partial, int rows, int d, int dt, void* stream);

/* ---- elementwise -------------------------------------------------------------------- */
/* LlamaMLP: out[t, f] = silu(gu[t, f]) * gu[t, ff + f]   (gu = fused gate|up GEMM output) */
int mh_swiglu_fwd(const void* gu, void* out, int rows, int ff, int dt, void* stream);
/* dgu[t, :] from dout; overwrites dgu */
int mh_swiglu_bwd(const void* gu, const void* dout, void* dgu, int rows, int ff, int dt, void* stream);
int mh_quick_gelu_fwd(const void* x, void* y, int64_t n, int dt, void* stream);
int mh_quick_gelu_bwd(const void* x, const void* dy, void* dx, int64_t n, int dt, void* stream);
/* y = a + b (n elements, 16-bit) */
int mh_add(const void* a, const void* b, void* y, int64_t n, int dt, void* stream);
/* generic dtype conversion src(dt_src) -> dst(dt_dst), n elements */
int mh_convert(const void* src, int dt_src, void* dst, int dt_dst, int64_t n, void* stream);

/* ---- RoPE (rotate-half, theta) on the fused qkv buffer [T, 3, H, D] in place -------------
 * llama_flash_attn_monkey_patch.py:56-59 (apply_rotary_pos_emb).  pos = t % S.  inverse=1
 * applies the transposed rotation (backward). */
int mh_rope_table(float* cos_sin, int S, int D, float theta, void* stream); /* [S, D/2, 2] */
int mh_rope_qk(void* qkv, const float* cos_sin, int T, int S, int H, int D, int inverse, int dt, void* stream);

/* ---- attention ---------------------------------------------------------------------- */
/* Flash attention forward.  q/k/v are [B*S, H, D] views with row strides ldq/ldk/ldv (elements): V is taken ROW-MAJOR and
 * transposed on the fly by LDS transpose-reads; o is [B*S, H*D] (ldo); lse fp32 [B, H, S_pad], S_pad = round_up(S, 64).
 * seqlens int32[B] or null (= S): keys >= seqlens[b] are excluded and query rows >= seqlens[b] are written as zeros,
 * i.e. flash_attn_varlen + pad_input semantics (llama_flash_attn_monkey_patch.py:87-102).
 * causal=1: Llama (D=128), causal=0: CLIP (D=64).  scale = 1/sqrt(D). */
int mh_attn_fwd2(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo,
                 float* lse, const int32_t* seqlens, int B, int S, int H, int D, int causal, int dt, void* stream);
/* D = 128: 1 = the forward runs in its PING-PONG form (csrc/attn_fwd3.hip: 8-wave 256-query blocks, the two waves of a SIMD held in
 * opposite phases - one in its MFMA section while the other is in its softmax section), 2 = ONE WAVE PER SIMD with 64 query rows per
 * wave (csrc/attn_fwd4.hip: K fragments resident in AccVGPRs, the softmax of one 32-row half between the MFMAs of the other),
 * 0 (default) = two free-running 128-query blocks per CU (attn_fwd2.hip).  A-B switch; bit-identical results
 * (profiles/r03_attn_fwd_pingpong.txt, profiles/r04_attn_fwd_wave64.txt). */
void mh_attn_fwd_pingpong(int on);
/* Backward: dq/dk/dv are [B*S, H, D] views with their own row strides; no workspace and no operand re-layout passes
 * (transposed operands come from LDS transpose-reads); rope_cos_sin != NULL applies the inverse RoPE to dq, dk.  `delta`: ZERO-INITIALISED fp32 scratch of 2*B*H*S_pad floats. */
int mh_attn_bwd2(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o, int64_t ldo,
                 const void* dout, int64_t lddo, const float* lse, float* delta, void* dq, int64_t lddq, void* dk, int64_t lddk,
                 void* dv, int64_t lddv, const int32_t* seqlens, int B, int S, int H, int D, int causal, const float* rope_cos_sin, int dt, void* stream);
/* The same backward in its FIVE-product form (D = 128, causal, S % 128 == 0, seqlens == NULL; any other case runs exactly mh_attn_bwd2): the
 * dK|dV kernel writes the unscaled dS = P o (dP - delta) it computes anyway (16-bit, causal half only) into `ds_ws`
 * (mh_attn_bwd_spill_bytes(B, S, H) bytes, scratch: may be shared by all layers of a step) and dQ = dS K is a one-product pass over it -
 * no second evaluation of the scores, dP and the exponentials (flash-attention's backward executes 7 products for 5; the chip is power-capped,
 * so executed work is what a kernel is billed for: profiles/r05_attn_bwd_spill.txt).  dK, dV are bit-identical to mh_attn_bwd2's, dQ agrees to
 * rounding (P comes from the dK|dV kernel's exp2(s * c) with the accumulator started at -lse / scale instead of exp2(fma(s, c, -lse * log2e))). */
int64_t mh_attn_bwd_spill_bytes(int B, int S, int H);
int mh_attn_bwd2_spill(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o, int64_t ldo,
                       const void* dout, int64_t lddo, const float* lse, float* delta, void* dq, int64_t lddq, void* dk, int64_t lddk,
                       void* dv, int64_t lddv, const int32_t* seqlens, int B, int S, int H, int D, int causal, const float* rope_cos_sin, int dt,
                       void* ds_ws, void* stream);
/* dK and dV at D = 128: 2 (default) = attn_bwd3_kv_k (one kernel; register-staged tile copies, three LDS stages, one barrier in the middle
 * of a tile), 1 = attn_bwd2_kv_k<MODE 3> (one kernel, LDS-DMA copies: rounds 2-4), 0 = two kernels.  A/B switch, bit-identical results. */
void mh_attn_bwd_fused_kv(int on);
/* The mode in force (0 / 1 / 2).  mh_attn_bwd2_spill takes its five-product form ONLY in mode 2 (the dS spill lives in attn_bwd3_kv_k); in modes
 * 0 / 1 it runs exactly mh_attn_bwd2 and never touches ds_ws - callers that report which form ran ask here.  Env MH_ATTN_BWD_FUSED_KV sets the
 * mode at import (since round 5 the value 1 selects the round 2-4 one-kernel form; before that it was the default and a no-op). */
int mh_attn_bwd_fused_kv_mode(void);
/* The row-per-lane epilogues of the attention kernels (o; dq, dk, dv) write 16 bytes per lane after a half-wave exchange (default, needs
 * 16-byte aligned rows: ld % 8 == 0) instead of 8 (0): A/B switch, bit-identical results (profiles/r05_attn_wide_stores.txt).
 * Limits of mh_attn_fwd2 / mh_attn_bwd2: one batch element's rows of q / k / v / dout must span < 2^31 bytes (S * ld * 2 < 2^31: the
 * tile copies use a 31-bit buffer-descriptor range; MH_ERR_SHAPE otherwise - with fused q|k|v rows of 3 * 4096 channels that is S < 87 381). */
void mh_attn_wide_stores(int on);
/* ---- KV-cache decode (S_q = 1): generate() behind llama_mmgpt.py:114-134 / eval_mmvet.py:101-120 ------------------
 * HF LlamaAttention with past_key_values (modeling_llama.py:243-281): q,k of the new token rotated at its own
 * position, k,v appended to the cache, softmax(q K^T / sqrt(D)) V over keys [0, len).  All HBM-bound kernels. */
/* y[m, n] = sum_k x[m, k] W[n, k] (+ resid[m, n]); M <= 8 activation rows against a weight streamed once; out is `dt`
 * or fp32.  K % 8 == 0, 16-byte aligned rows. */
int mh_gemv(const void* x, int64_t ldx, const void* W, int64_t ldw, void* out, int64_t ldo, const void* resid, int64_t ldr,
            int M, int N, int K, int dt, int out_f32, void* stream);
/* fp8 weight path of the decode step (BASELINE cfg 5's weight format: OCP e4m3 values, one fp32 scale per 128
 * consecutive k of a row; activations stay `dt`).  mh_quant_fp8_b128: q[N, K] bytes + scales[N, ceil(K/128)] from
 * w[N, K] (`dt`).  mh_gemv_fp8w: as mh_gemv with the weight given as (q, scales); K % 16 == 0. */
int mh_quant_fp8_b128(const void* w, int64_t ldw, void* q, float* scales, int N, int K, int dt, void* stream);
int mh_gemv_fp8w(const void* x, int64_t ldx, const void* q, const float* scales, void* out, int64_t ldo, const void* resid,
                 int64_t ldr, int M, int N, int K, int dt, int out_f32, void* stream);
/* mh_gemv / mh_gemv_fp8w take up to 16 activation rows; from 3 rows on they run as an MFMA kernel (16-64 weight rows per block
 * streamed straight into the B-operand registers, activations straight into the A operand, K split over 8-16 waves) instead of one
 * wave per weight row.  A/B switch (sets the threshold for both weight formats; 17 = never, <= 0 = default): */
void mh_gemv_mfma_min_rows(int rows);
/* Row count from which the decode projections with a paired epilogue (mh_gemv_swiglu, mh_gemv_qkv_rope, mh_gemv_fp8w_norm with ff > 0)
 * use the MFMA form: 16-bit weights (default 6) and fp8 weights (default 4); <= 0 = default. */
void mh_gemv_mfma_pair_min_rows(int rows16, int rows_fp8);
/* 1-2 rows, N <= 8192 (o / down projections): 1 (default) = the four waves of a block split K (4x the waves), 0 = one wave per row pair. */
void mh_gemv_ksplit(int on);
/* MFMA form at N <= 8192: 1 (default) = 16 waves per block split K, 0 = 8. */
void mh_gemv_mfma_wide(int on);
/* Decode-step MLP gate|up projection + SwiGLU in one launch (HF LlamaMLP, modeling_llama.py:174-176, one token per sequence):
 * act[M, ff] = silu(x Wg^T) * (x Wu^T), Wgu = [Wg; Wu] [2 ff, K] row-major; gate / up are rounded to 16 bits before the
 * activation exactly as mh_gemv + mh_swiglu_fwd do (same result up to the last bit of the activation).  M <= 8. */
int mh_gemv_swiglu(const void* x, int64_t ldx, const void* Wgu, int64_t ldw, void* act, int64_t ldo, int M, int ff, int K, int dt,
                   void* stream);
/* Decode step: the RMSNorm in front of the q|k|v or gate|up projection folded into the projection's launch (HF LlamaDecoderLayer,
 * modeling_llama.py input_layernorm / post_attention_layernorm): out[M, N] = rmsnorm(x; norm_w, eps) W^T; with ff > 0, W = [Wg; Wu] and
 * out[M, ff] = silu(gate) * up as mh_gemv_swiglu.  Every block normalises the rows itself (no separate norm kernel).  M <= 8, K <= 8192. */
int mh_gemv_norm(const void* x, int64_t ldx, const void* norm_w, float eps, const void* W, int64_t ldw, void* out, int64_t ldo, int M, int N,
                 int K, int ff, int dt, void* stream);
/* The same with fp8 (OCP e4m3, one fp32 scale per 128 k: mh_quant_fp8_b128) weights; norm_w == NULL: no norm (x is used as it is). */
int mh_gemv_fp8w_norm(const void* x, int64_t ldx, const void* norm_w, float eps, const void* q, const float* scales, void* out, int64_t ldo,
                      int M, int N, int K, int ff, int dt, void* stream);
/* The q|k|v projection of one decode step in ONE launch: optional input_layernorm (norm_w, may be NULL), projection with 16-bit weights W
 * [3 H D, K] or (W == NULL) fp8 weights q8 + scales, rotate-half RoPE of q and k at pos[m] (llama_flash_attn_monkey_patch.py:56-59 on one
 * token) and the append of k, v to kcache / vcache [M, Smax, H D] at row pos[m].  qkv [M, 3 H D] gets the rotated q, k and v.  Equal to
 * mh_rmsnorm_fwd + mh_gemv (mh_gemv_fp8w) + mh_decode_rope_append bit for bit.  M <= 8, K <= 8192.
 * rope_pos (NULL = pos): rotary position of the new token when it differs from its cache row - prompts with padding in front of or
 * inside them keep only their valid keys in the cache (rows 0..count-1) while positions stay absolute, as HF's LlamaModel numbers them
 * when no position_ids are passed (llama_mmgpt.py:114-134 passes none). */
int mh_gemv_qkv_rope(const void* x, int64_t ldx, const void* norm_w, float eps, const void* W, int64_t ldw, const void* q8, const float* scales,
                     void* qkv, int64_t ldo, int M, int K, int dt, const float* cos_sin, const int32_t* pos, const int32_t* rope_pos,
                     void* kcache, void* vcache, int H, int D, int Smax, void* stream);
/* qkv [B, 3, H, D] of the new tokens: rotate q and k in place at position rope_pos[b] (int32, device; NULL = pos), copy k and v into
 * kcache / vcache [B, Smax, H*D] at row pos[b]. */
int mh_decode_rope_append(void* qkv, const float* cos_sin, const int32_t* pos, const int32_t* rope_pos, void* kcache, void* vcache, int B,
                          int H, int D, int Smax, int dt, void* stream);
/* out[b, h*D..] = softmax(q[b,h] . K[b, 0..lens[b]) / sqrt(D)) V[b, 0..lens[b]);  q row stride ldq; D in {64, 128}.
 * Split-KV: with ws != NULL (B*H*mh_attn_decode_splits(B,H,Smax)*(D+2) floats) several blocks share one (b, h) and a second
 * launch merges their partial softmaxes; ws == NULL runs one block per (b, h).  mh_attn_decode_fused_merge(1): the last block of a
 * (b, h) to finish merges instead (ticket counters inside the library: one decode stream at a time; A-B arm, default off). */
int mh_attn_decode_splits(int B, int H, int Smax);
void mh_attn_decode_fused_merge(int on);
int mh_attn_decode(const void* q, int64_t ldq, const void* kcache, const void* vcache, void* out, const int32_t* lens,
                   int B, int H, int D, int Smax, float* ws, int dt, void* stream);

/* ---- token selection for generate() (HF GenerationMixin as the reference's eval scripts drive it: eval_mmvet.py:101-120
 * `do_sample=True, temperature=0.2` or `num_beams=5`; site-packages transformers/generation/logits_process.py) ------------
 * logits fp32 [rows, ldl], V valid columns.  do_sample=0: out[r] = lowest index of the row maximum (greedy).
 * do_sample=1: TemperatureLogitsWarper (z = logits / temperature) -> TopKLogitsWarper (top_k > 0: keep z >= k-th largest, ties
 * kept; 0 = off) -> TopPLogitsWarper (top_p < 1: drop tokens whose ascending cumulative probability is <= 1 - top_p, keep >= 1)
 * -> multinomial over the remaining softmax by inverse CDF in index order.  The uniform of row r at decoding step `step` is
 * u = (splitmix64(seed ^ splitmix64(step * 0x100000001B3 + r)) >> 40) * 2^-24 (counter-based: nothing comes from the host,
 * the same (seed, step, row) always draws the same token); out_u (nullable) receives it.  out int64 [rows]. */
int mh_select_tokens(const float* logits, int64_t ldl, int rows, int V, int do_sample, float temperature, int top_k, float top_p,
                     uint64_t seed, int64_t step, int64_t* out, float* out_u, void* stream);
/* out[r, :V] = log_softmax(logits[r, :V]) + row_bias[r] (fp32; row_bias nullable): beam search's accumulated scores
 * log_probs + running_beam_scores (transformers/generation/utils.py `_beam_search`) */
int mh_log_softmax_rows(const float* logits, int64_t ldl, int rows, int V, float* out, int64_t ldo, const float* row_bias, void* stream);
/* dst[i, :cols_bytes] = src[idx[i], :cols_bytes], zeros where idx[i] < 0; row strides and cols in BYTES, multiples of 4 (16 for the fast path) (KV-cache reorder by
 * beam index = HF `_reorder_cache`; expansion of a prefilled batch to num_beams rows per prompt) */
int mh_gather_rows2d(const void* src, int64_t lds_bytes, const int64_t* idx, void* dst, int64_t ldd_bytes, int rows, int64_t cols_bytes,
                     void* stream);

/* ---- fp32-store parity mode (SURVEY §8d cfg 2; not a performance path): the forward with every activation held in fp32, for
 * comparing the kernels' arithmetic with the reference's fp32 CPU path at BASELINE's "1e-3 rel" without the rounding of 16-bit
 * activation storage.  Linear layers: x (fp32) is split exactly into three bf16 terms (mh_p32_split3) and multiplied on the
 * production bf16 MFMA GEMM (mh_gemm with MH_EPI_OUT_F32 | MH_EPI_ACCUM), weights being exactly representable in bf16; the
 * remaining ops are plain fp32 kernels with bf16 parameters: HF LlamaRMSNorm / CLIP LayerNorm, rotate-half RoPE
 * (llama_flash_attn_monkey_patch.py:56-59), SwiGLU (op 1) / quick-GELU (op 2) / add (op 0), embedding + splice
 * (base_mmgpt.py:99-160), patch im2col + CLS/position assembly, conv-projector gather, causal / key-padded attention
 * (one wave per query row, online softmax; semantics of mh_attn_fwd2). */
int mh_p32_split3(const float* x, void* hi_bf16, void* mid_bf16, void* lo_bf16, int64_t n, void* stream);
int mh_p32_rmsnorm(const float* x, const void* w_bf16, float* y, int rows, int d, float eps, void* stream);
int mh_p32_layernorm(const float* x, const void* w_bf16, const void* b_bf16, float* y, int rows, int d, float eps, void* stream);
int mh_p32_elementwise(const float* a, const float* b, float* y, int64_t n, int op, int ff, void* stream);
int mh_p32_rope(float* qkv, const float* cos_sin, int64_t T, int S, int H, int D, void* stream);
int mh_p32_embed_splice(const int64_t* ids, const int32_t* src, const void* embed_bf16, const float* feats, float* out, int64_t T, int d, void* stream);
int mh_p32_im2col(const float* pixels, float* cols, int N, int img, int ps, int Kpad, int rows_per_img, int row0, void* stream);
int mh_p32_vit_assemble(const float* patch, const void* cls_bf16, const void* pos_bf16, float* x, int N, int G2, int d, void* stream);
int mh_p32_conv3x3_cols(const float* x, float* cols, int N, int G, int C, int stride, int rows_per_img, int row0, void* stream);
int mh_p32_attention(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, float* o, int64_t ldo,
                     const int32_t* seqlens, int B, int S, int H, int D, int causal, void* stream);

/* ---- CLIP patch embedding ------------------------------------------------------------- */
/* cols[n*rows_per_img + row0 + p, c*ps*ps + py*ps + px] = pixels[n, c, gy*ps+py, gx*ps+px], zero padded to Kpad; the
 * first row0 rows of every image (the CLS slot when rows_per_img = G*G+1, row0 = 1) are zero, so the token-major layout
 * of the tower is used from the first GEMM on and the patch-embedding weight gradient is ONE K-strided GEMM over all
 * rows.  pixels fp32 (pix_dt = MH_F32) or 16-bit; cols `dt`.  clip_encoder.py:76-79 -> CLIPVisionEmbeddings. */
int mh_im2col_patches(const void* pixels, int pix_dt, void* cols, int N, int img, int ps, int Kpad, int rows_per_img, int row0,
                      int dt, void* stream);
/* x[n, 0, :] = cls + pos[0];  x[n, 1+p, :] = patch[n*(G2+1) + 1 + p, :] + pos[1+p]   (then pre_layrnorm) */
int mh_vit_assemble(const void* patch, const void* cls, const void* pos, void* x, int N, int G2, int d, int dt, void* stream);
/* dst[r, c] (=|+=) src[r, c], r < rows, c < cols: strided 2-D block copy / accumulate of 16-bit data (pads the
 * patch-embedding weight's K = 588 to 640 once per weight version and un-pads its gradient) */
int mh_copy2d(const void* src, int64_t lds, void* dst, int64_t ldd, int rows, int cols, int accumulate, int dt, void* stream);
/* backward of assemble: dpatch rows copy, dcls/dpos partial sums are taken with mh_colsum */

/* ---- ConvProjector (conv_projector.py:23-39): Conv2d(C->d, k3, stride s, pad 1) as an implicit GEMM -------------
 * cols[(n,oy,ox), c*9 + ky*3 + kx] gathered from x [N*rows_per_img, C] (patch p of image n at row
 * n*rows_per_img + row0 + p, zero padding outside the G x G grid); then mh_gemm with weight.view(d, C*9). */
int mh_conv3x3_cols(const void* x, void* cols, int N, int G, int C, int stride, int rows_per_img, int row0, void* stream);
/* backward of the gather: dx[N*rows_per_img, C] from dcols (deterministic gather form; non-patch rows = 0) */
int mh_conv3x3_col2im(const void* dcols, void* dx, int N, int G, int C, int stride, int rows_per_img, int row0, int dt, void* stream);

/* ---- embedding + image-feature splice (base_mmgpt.py:99-160) ------------------------------ */
/* Builds src[b*S+s] = row index into the image-feature matrix [Nimg*P, d] if position s of
 * sample b is one of the P rows after an <im_start>, else -1 (use embed_tokens[ids]).
 * img_offset int32[B+1]: exclusive prefix sum of images per sample.  err int32[4] (device):
 * err[0] != 0 on <im_start>/<im_end> count mismatch (base_mmgpt.py:116-118), err[1] != 0 when
 * <im_end> is not at start+P+1 (base_mmgpt.py:125-126); err[2], err[3] = (sample, position).
 * Feature row of patch j of global image g is g*rows_per_img + row0 + j (rows_per_img=577,row0=1
 * lets the projector run on the tower output with its CLS rows in place: no drop-CLS copy). */
int mh_splice_index(const int64_t* ids, const int32_t* img_offset, int32_t* src, int32_t* err,
                    int B, int S, int P, int64_t im_patch, int64_t im_start, int64_t im_end,
                    int rows_per_img, int row0, void* stream);
/* lens[b] = 1 + last s with mask[b, s] != 0 (mask: bool/uint8 [B, S]); the collator's right-padded
 * attention_mask (collator.py:32) -> per-sample lengths = flash-attn's cu_seqlens */
int mh_mask_lens(const void* mask_u8, int32_t* lens, int B, int S, void* stream);
/* unpad_input / pad_input of the key-padding branch (llama_flash_attn_monkey_patch.py:87-102; flash_attn.bert_padding) as row tables for
 * mh_gather_rows2d: fwd[b*S + r] = flat row of the r-th valid position of sample b (-1 for r >= count[b]); inv[b*S + s] = b*S + rank of
 * position s among the valid ones (-1 where mask is 0).  Valid tokens are compacted to the front of their own sample's S rows, so
 * the attention kernels see a right-padded batch with lens = count: ANY mask (left padding, holes) runs, like the reference. */
int mh_mask_unpad_index(const void* mask_u8, int64_t* fwd, int64_t* inv, int32_t* count, int B, int S, void* stream);
/* Device-side validation of a batch (no host sync; the caller reads `err` back asynchronously):
 *   err[4]/err[5]: an input id outside [0, V) / its flat position      (reference: torch embedding raises IndexError)
 *   err[6]/err[7]: a label that is neither -100 nor in [0, V) / position (reference: CrossEntropyLoss target out of bounds)
 *   err[8]/err[9]: attention_mask of sample err[9] is not a right-padded prefix (popcount != lens[b]): not an error - the host then
 *   routes attention through mh_mask_unpad_index's tables (general key-padding form of llama_flash_attn_monkey_patch.py:87-102)
 *   instead of the lens-only fast path of right-padded batches (collator.py:29-34).
 *   err[10]: some sample's mask holds a zero (the batch carries padding).  0 = every sequence is S long: the host then drops the lengths
 *   and the attention kernels run their no-lengths forms (no masks in the tile loops; backward as five products, mh_attn_bwd2_spill).
 * ids / labels / mask may each be NULL (skipped).  err is int32[12], shared with mh_splice_index (slots 0-3). */
int mh_check_inputs(const int64_t* ids, const int64_t* labels, const void* mask_u8, const int32_t* lens, int32_t* err, int B, int S,
                    int V, void* stream);
int mh_embed_splice_fwd(const int64_t* ids, const int32_t* src, const void* embed, const void* feats,
                        void* out, int T, int d, int dt, void* stream);
/* dfeats[src] = dout rows (pure copy; rows never collide); dembed32[ids] += dout (fp32 atomics) */
int mh_embed_splice_bwd(const int64_t* ids, const int32_t* src, const void* dout, void* dfeats, float* dembed32,
                        int T, int d, int dt, void* stream);

/* ---- shifted cross-entropy (llama_mmgpt.py:92-100) -------------------------------------- */
/* logits fp32 [B*S, ldl]; labels int64 [B, S].  Row (b,s) is scored against labels[b, s+1]
 * (ignored when s == S-1 or label == -100).  Writes row_loss[T] (0 for ignored), lse[T];
 * out[0] = sum of losses, out[1] = number of scored rows, out[2] = mean (fp32[4]), via a final
 * 1-block reduce (deterministic). */
int mh_ce_fwd(const float* logits, int64_t ldl, const int64_t* labels, float* row_loss, float* lse, float* out2,
              int B, int S, int V, void* stream);
/* dlogits[t, v] (dt, ld = lddl, zero for v in [V, Vpad)) = gscale/count * (softmax - onehot) */
int mh_ce_bwd(const float* logits, int64_t ldl, const int64_t* labels, const float* lse, const float* out2,
              void* dlogits, int64_t lddl, int B, int S, int V, int Vpad, float gscale, int dt, void* stream);
/* The same gradient in COMPACT form: output row r = the gradient of logits row rows[r] (flat position; rows[r] < 0: a zero row), nrows rows.
 * The reference scores only positions whose shifted label is not -100 (llama_mmgpt.py:92-100); every other row of dlogits is exactly zero, so the
 * head's dgrad and wgrad contract over the scored rows alone (cfg 3: 603 of 4096 positions per sequence) - engine.backward, `sparse_head`. */
int mh_ce_bwd_rows(const float* logits, int64_t ldl, const int64_t* labels, const float* lse, const float* out2, void* dlogits, int64_t lddl,
                   const int64_t* rows, int nrows, int S, int V, int Vpad, float gscale, int dt, void* stream);

/* ---- optimizer (reference: torch AdamW via HF Trainer, trainer.py:45-74) ------------------- */
/* p, g are `dt`; m, v fp32.  Decoupled weight decay, bias-corrected; gscale multiplies g (1/world, clip). */
int mh_adamw(void* p, const void* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
             float wd, int step, float gscale, int dt, void* stream);
/* Same update with an extra DEVICE-side gradient factor (gscale * *gscale_dev): global-norm clipping without a host
 * round trip.  mh_clip_scale: out2[0] = min(1, max_norm / (sqrt(*sumsq) * |gscale| + 1e-6)), out2[1] = that norm
 * (torch.nn.utils.clip_grad_norm_ as HF Trainer applies it with --max_grad_norm; trainer.py:45-74 builds the optimizer). */
int mh_adamw_clip(void* p, const void* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                  float wd, int step, float gscale, const float* gscale_dev, int dt, void* stream);
int mh_clip_scale(const float* sumsq, float gscale, float max_norm, float* out2, void* stream);
/* out[0] += sum(g^2) over n elements (fp32 atomic; zero it first) */
int mh_sumsq(const void* g, int64_t n, float* out, int dt, void* stream);
/* Deterministic form (fixed grid, fixed-order two-stage sum; 16-byte loads): out[0] = sum g^2; `partial` = 2048 floats
 * of scratch.  Used for gradient clipping so that the clip coefficient - and therefore training - is run-to-run reproducible. */
int mh_sumsq_det(const void* g, int64_t n, float* partial, float* out, int dt, void* stream);
/* *flag |= 1 when any of the n 16-bit elements (bf16 or fp16, 16-byte aligned) has a non-zero magnitude: an exact test on the bit patterns
 * (-0 counts as zero; NaN, Inf and values whose square underflows fp32 do not).  merlin_amd/dp.py uses it before it lets a backward
 * accumulate onto an attached gradient arena that was already all-reduced (`zero_grad(set_to_none=False)` workflows). */
int mh_any_nonzero(const void* g, int64_t n, int* flag, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MERLIN_HIP_H */
