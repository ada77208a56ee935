"""The hot path: MMGPTLlamaForCausalLM forward + backward as a hand-scheduled sequence of HIP
kernels (include/merlin_hip.h) over the flat parameter arena.  No autograd graph, no torch math:
torch supplies device memory and the stream.

Forward (reference call chain -> here):
  llama_mmgpt.py:72   prepare_inputs_labels_for_multimodal -> tower() + projector() + splice()
  clip_encoder.py:74-82 / HF CLIPVisionModel            -> tower()   (im2col+GEMM, LN, 23 layers)
  mlp_projector.py:19-23 / conv_projector.py:23-39      -> projector()
  base_mmgpt.py:99-160                                  -> splice()  (index kernel + one gather)
  llama_mmgpt.py:75   LlamaModel (32 x decoder layer)   -> llama_layer_fwd()
  llama_flash_attn_monkey_patch.py:20-103               -> fused QKV GEMM + RoPE + flash attention
  llama_mmgpt.py:87-100 lm_head + shifted CE            -> head_loss()
Backward mirrors it layer by layer.  Each decoder / encoder layer keeps only its input; the layer
forward is recomputed inside the backward (the reference trains with --gradient_checkpointing True,
pretrain.sh:31), or, with `save_activations=True`, kept resident (288 GB HBM makes that affordable).
Weight gradients are written straight into the gradient arena (fused accumulate in the GEMM
epilogue); `on_grads_ready(name_list)` fires per layer so the data-parallel all-reduce of that bucket
overlaps the rest of the backward (merlin_amd/dp.py).
"""
from __future__ import annotations

import math
import os
import weakref

import torch

from .. import ops as O
from .arena import Arena
from .config import head_dim_of, rope_theta_of

VT = "model.vision_tower.vision_tower.vision_model."


def _ru(x, m):
    return (x + m - 1) // m * m


class _LlamaLayerW:
    __slots__ = ("wqkv", "wo", "wgu", "wd", "ln1", "ln2", "names", "p")


class _VitLayerW:
    __slots__ = ("wqkv", "bqkv", "wo", "bo", "w1", "b1", "w2", "b2", "ln1w", "ln1b", "ln2w", "ln2b", "names", "p")


class HipEngine:
    def __init__(self, model):
        self._model = weakref.ref(model)
        self.arena = None
        self._arena_sig = None
        self.rope = None
        self.save_activations = False
        self.on_grads_ready = None  # callable(list_of_param_names) | None: a bucket's gradients are final (dp.GradSync)
        self.on_backward_begin = None  # callable(fresh: bool) | None
        self.strict_checks = True
        self.parity_fp32 = False  # opt-in checking mode: fp32-store forward (merlin_amd/parity.py)
        self.keep_full_lengths = False  # tests: pass the per-sample lengths to the attention kernels even when every sequence fills the context
        self.force_unpad = False  # tests: send right-padded masks through the general unpad / pad attention path as well
        # fp8 training step beyond the decoder: lm_head on by default (-0.5 % of a cfg-5 step); the CLIP tower's Linears (K = 1024: their
        # quantisation passes cost more than the fp8 MFMA returns, +0.6 %, profiles/r03_fp8_parts_ab.txt) implemented, tested, off by default
        self.fp8_head = True
        self.fp8_tower = False
        # fp8 step: the SwiGLU-backward dgrad takes the row / column maxima of dgu in its store phase (mh_gemm_fp8_swiglu_bwd_amax) so that the quantiser
        # behind it reads the layer's largest gradient tensor once instead of twice; identical bytes (tests/test_fp8_training_gpu.py); env MH_FP8_FUSED_AMAX=0: A/B
        self.fp8_fused_amax = os.environ.get("MH_FP8_FUSED_AMAX", "1") != "0"
        # Both towers' residual streams are fp32 (updated in place by the accumulating fp32 epilogue of the projections that feed them,
        # read by mh_norm_fwd_f32in); GEMM operands, attention and every saved activation stay 16-bit, the backward runs on the 16-bit
        # copies of the layer inputs the stream's reader emits.  Default ON since round 4: the configuration that is benchmarked is the
        # one whose parity is quoted (full-depth 7B fp16 logits 4.6e-3 -> 2.1e-3 of max|logit| against the reference's fp32 output, within
        # 1.1x of the measured 16-bit-operand floor: tests/test_parity_floor_gpu.py) at +1.4 % step time.  False = 16-bit streams (the
        # reference's own bf16 training numerics).  The fp8 paths keep 16-bit streams.
        self.fp32_residual = True
        # Resident-activation footprint (save_activations = True), a slope instead of the cliff to full layer recompute (bench.py picks the lowest
        # level that leaves --min-free-gb of HBM next to RCCL's buffers): 0 = keep everything (251 GB at cfg 3, the fastest); 1 = do not keep the
        # normed GEMM operands h1 / h2 of the decoder and tower layers - the backward re-derives them from the saved 16-bit layer inputs with the
        # forward's norm kernels right before the weight gradients that read them (-19.8 GB for ~7 ms per step); 2 = additionally recompute
        # act = silu(gate) * up from the saved gate|up tensor in the first `mem_act_layers` decoder layers (-0.72 GB and +0.45 ms per layer).
        self.mem_level = 0
        self.mem_act_layers = 16
        # lm_head backward over the SCORED rows only: the reference's loss ignores every position whose shifted label is -100
        # (llama_mmgpt.py:92-100), so those rows of dlogits are exactly zero and contribute nothing to d(hidden) = dlogits W and
        # dW = dlogits^T hidden.  The scored rows (cfg 3: the 603 trajectory positions of each 4096-position sequence) are compacted on the
        # device (mh_mask_unpad_index over the flat batch; their count is read back without a wait) and both products contract over them alone
        # whenever they are at most half of the batch.  Exact arithmetic, fewer zero terms; False = the dense products.
        self.sparse_head = os.environ.get("MH_DENSE_HEAD", "0") != "1"  # (env: A/B switch for benchmarks)
        self.sparse_last_layer = os.environ.get("MH_DENSE_LAST_LAYER", "0") != "1"  # last decoder layer's MLP / o-proj backward over the scored rows
        self._err = None
        self.weight_version = 0  # bumped whenever parameter VALUES change (optimizer step, loads, repack): derived copies
        self._derived = {}       # (fp8 weights, the K-padded patch-embedding weight) are keyed on it
        self._pver = None        # torch version counters of the parameters at the last forward (ensure_arena)

    # ------------------------------------------------------------------------------------------
    # parameter arena
    # ------------------------------------------------------------------------------------------
    @property
    def model(self):
        return self._model()

    def _ordered_params(self):
        m = self.model
        named = dict(m.named_parameters())
        order = []
        cfg = m.config
        for i in range(cfg.num_hidden_layers):
            p = f"model.layers.{i}."
            order += [p + "self_attn.q_proj.weight", p + "self_attn.k_proj.weight", p + "self_attn.v_proj.weight",
                      p + "self_attn.o_proj.weight", p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight",
                      p + "mlp.down_proj.weight", p + "input_layernorm.weight", p + "post_attention_layernorm.weight"]
        order += ["model.norm.weight", "lm_head.weight", "model.embed_tokens.weight"]  # (norm, lm_head) = one DP bucket
        if any(k.startswith(VT) for k in named):
            vt = m.get_model().vision_tower
            for i in range(vt.config.num_hidden_layers):
                p = VT + f"encoder.layers.{i}."
                order += [p + f"self_attn.{n}_proj.weight" for n in ("q", "k", "v")]
                order += [p + f"self_attn.{n}_proj.bias" for n in ("q", "k", "v")]
                order += [p + "self_attn.out_proj.weight", p + "self_attn.out_proj.bias",
                          p + "layer_norm1.weight", p + "layer_norm1.bias", p + "layer_norm2.weight", p + "layer_norm2.bias",
                          p + "mlp.fc1.weight", p + "mlp.fc1.bias", p + "mlp.fc2.weight", p + "mlp.fc2.bias"]
            order += [VT + "embeddings.class_embedding", VT + "embeddings.patch_embedding.weight",
                      VT + "embeddings.position_embedding.weight", VT + "pre_layrnorm.weight", VT + "pre_layrnorm.bias",
                      VT + "post_layernorm.weight", VT + "post_layernorm.bias"]
        rest = [k for k in named if k not in set(order)]
        order += rest
        missing = [k for k in order if k not in named]
        if missing:
            raise RuntimeError(f"parameters expected by the HIP engine are missing: {missing[:4]}...")
        return [(k, named[k]) for k in order]

    def ensure_arena(self):
        m = self.model
        sig = tuple(id(p) for _, p in m.named_parameters())
        if self.arena is None or self._arena_sig != sig:
            cfg = m.config
            vpad = _ru(cfg.vocab_size, 64)
            self.arena = Arena(self._ordered_params(), alloc_numel={"lm_head.weight": vpad * cfg.hidden_size})
            self._arena_sig = sig
            self._bind = None
        if self.arena.ensure_packed() or self._bind is None:
            self._bind_views()
        # Parameters are views of the arena, so ANY in-place update of one (a stock torch.optim step on param.grad under HF
        # Trainer, p.mul_(), clip-by-value, EMA swaps ...) changes the arena behind the derived copies.  torch counts in-place
        # writes per tensor (`_version`): a changed signature drops the copies exactly like FusedAdamW.step / load_state_dict do.
        # (Writes through `p.data` get a fresh version counter from torch and stay invisible: call weights_changed() after those.)
        pv = self._param_versions()
        if pv != self._pver:
            if self._pver is not None:
                self.weights_changed()
            self._pver = pv
        return self.arena

    def _param_versions(self):
        return tuple(p._version for p in self.arena.params.values())

    def _bind_views(self):
        """Fused weight views (q|k|v, gate|up) over the arena."""
        m, A = self.model, self.arena
        cfg = m.config
        d, ff = cfg.hidden_size, cfg.intermediate_size
        self.llama = []
        for i in range(cfg.num_hidden_layers):
            p = f"model.layers.{i}."
            W = _LlamaLayerW()
            W.p = p
            W.wqkv = A.span(p + "self_attn.q_proj.weight", p + "self_attn.v_proj.weight", (3 * d, d))
            W.wo = A.view(p + "self_attn.o_proj.weight", shape=(d, d))
            W.wgu = A.span(p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight", (2 * ff, d))
            W.wd = A.view(p + "mlp.down_proj.weight", shape=(d, ff))
            W.ln1 = A.view(p + "input_layernorm.weight")
            W.ln2 = A.view(p + "post_attention_layernorm.weight")
            W.names = [n for n in A.names if n.startswith(p)]
            self.llama.append(W)
        self.vit = []
        tower = getattr(m.get_model(), "vision_tower", None)
        if tower is not None:
            vc = tower.config
            vd, vff = vc.hidden_size, vc.intermediate_size
            for i in range(vc.num_hidden_layers):
                p = VT + f"encoder.layers.{i}."
                W = _VitLayerW()
                W.p = p
                W.wqkv = A.span(p + "self_attn.q_proj.weight", p + "self_attn.v_proj.weight", (3 * vd, vd))
                W.bqkv = A.span(p + "self_attn.q_proj.bias", p + "self_attn.v_proj.bias", (3 * vd,))
                W.wo, W.bo = A.view(p + "self_attn.out_proj.weight", shape=(vd, vd)), A.view(p + "self_attn.out_proj.bias")
                W.w1, W.b1 = A.view(p + "mlp.fc1.weight", shape=(vff, vd)), A.view(p + "mlp.fc1.bias")
                W.w2, W.b2 = A.view(p + "mlp.fc2.weight", shape=(vd, vff)), A.view(p + "mlp.fc2.bias")
                W.ln1w, W.ln1b = A.view(p + "layer_norm1.weight"), A.view(p + "layer_norm1.bias")
                W.ln2w, W.ln2b = A.view(p + "layer_norm2.weight"), A.view(p + "layer_norm2.bias")
                W.names = [n for n in A.names if n.startswith(p)]
                self.vit.append(W)
        self._bind = True
        self.rope = None
        self.weights_changed()

    def weights_changed(self):
        """Parameter values changed (optimizer step, checkpoint load, init, repack): drop every derived copy."""
        self.weight_version += 1
        self._derived = {}
        self._fp8 = self._fp8_fwd = None

    def _derive(self, key, make):
        hit = self._derived.get(key)
        if hit is None or hit[0] != self.weight_version:
            hit = self._derived[key] = (self.weight_version, make())
        return hit[1]

    def _trainable(self, name):
        return self.arena.params[name].requires_grad

    # ------------------------------------------------------------------------------------------
    # helpers
    # ------------------------------------------------------------------------------------------
    def _rope_table(self, S, device):
        cfg = self.model.config
        if self.rope is None or self.rope.shape[0] < S or self.rope.device != device:
            # computed on the host exactly like transformers' LlamaRotaryEmbedding (fp32), then uploaded:
            # init-time plumbing; mh_rope_table is the device-side equivalent (tests compare them)
            D = head_dim_of(cfg)
            n = max(S, 64)
            inv = 1.0 / (rope_theta_of(cfg) ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
            fr = torch.outer(torch.arange(n, dtype=torch.float32), inv)
            self.rope = torch.stack((fr.cos(), fr.sin()), dim=-1).contiguous().to(device)
        return self.rope

    def _wgrad(self, dy, x, gout, fresh, Tpad):
        """gout[N_out, K_in] (+)= dy[T, N_out]^T @ x[T, K_in].  The contraction runs over tokens, so both operands
        are K-strided as they lie in memory: the MFMA kernel reads them with transpose-reads, no copies, any T
        (the CLIP tower's 27 696 tokens are not a multiple of the 64-token K-tile), split-K for the small CLIP
        weights.  Odd feature widths (not a multiple of 8) fall back to zero-padded transposed copies."""
        if dy.shape[1] % 8 == 0 and x.shape[1] % 8 == 0:
            O.wgrad_tn(dy, x, gout, accum=not fresh)
            return
        dyT = O.transpose16(dy, r_pad=Tpad)
        xT = O.transpose16(x, r_pad=Tpad)
        O.gemm_nt(dyT, xT, out=gout, accum=not fresh)

    def _ready(self, names):
        if self.on_grads_ready is not None:
            self.on_grads_ready(names)

    def _untouched(self, names, fresh):
        """A trainable bucket this backward does not reach: zero it on the first micro-step (the arena may hold the previous
        step's values), leave accumulated values alone otherwise, and report it like any other bucket."""
        if fresh:
            for n in names:
                if self.arena.params[n].requires_grad:
                    self.arena.gview(n).zero_()
        self._ready(names)

    # ------------------------------------------------------------------------------------------
    # vision tower
    # ------------------------------------------------------------------------------------------
    def _vit_layer_fwd(self, W, x, N, S, vc, keep):
        H = vc.num_attention_heads
        vd = vc.hidden_size
        D = vd // H
        eps = vc.layer_norm_eps
        h1 = O.layernorm_fwd(x, W.ln1w, W.ln1b, eps)
        qkv = O.gemm_nt(h1, W.wqkv, bias=W.bqkv)
        q, k, v = qkv[:, :vd], qkv[:, vd:2 * vd], qkv[:, 2 * vd:]
        o, lse = O.attn_fwd2(q, k, v, N, S, H, D, causal=False)
        x2 = O.gemm_nt(o, W.wo, bias=W.bo, resid=x)
        h2 = O.layernorm_fwd(x2, W.ln2w, W.ln2b, eps)
        if keep:
            f1, a = self._fc1_gelu(h2, W)  # f1 (kept for the backward) and quick_gelu(f1) from one launch
        else:
            f1 = None
            a = O.gemm_nt(h2, W.w1, bias=W.b1, act="quick_gelu")
        y = O.gemm_nt(a, W.w2, bias=W.b2, resid=x2)
        return y, (((None if self.mem_level else h1), qkv, o, lse, x2, (None if self.mem_level else h2), f1, a) if keep else None)

    @staticmethod
    def _fc1_gelu(h2, W):
        if W.w1.shape[0] % 8 == 0 and h2.shape[1] % 64 == 0:
            return O.gemm_gelu_fwd(h2, W.w1, W.b1)
        f1 = O.gemm_nt(h2, W.w1, bias=W.b1)
        return f1, O.quick_gelu_fwd(f1)

    def _vit_layer_bwd(self, W, x, dy, N, S, vc, saved, fresh):
        A = self.arena
        H = vc.num_attention_heads
        vd = vc.hidden_size
        D = vd // H
        eps = vc.layer_norm_eps
        if saved is None:
            _, saved = self._vit_layer_fwd(W, x, N, S, vc, keep=True)
        h1, qkv, o, lse, x2, h2, f1, a = saved
        if h1 is None:  # mem_level >= 1: re-derived from the saved 16-bit layer inputs
            h1 = O.layernorm_fwd(x, W.ln1w, W.ln1b, eps)
            h2 = O.layernorm_fwd(x2, W.ln2w, W.ln2b, eps)
        T = x.shape[0]
        Tpad = _ru(T, 64)
        p = W.p
        acc = not fresh
        # The four weight gradients of the layer are issued as ONE grouped launch (O.wgrad_tn_grouped: 48 + 16 + 64 + 64 output tiles of
        # 256^2 for ViT-L fill the chip together for the whole 27 696-token contraction; one by one each needed split-K with fp32
        # partials).  Its operands must therefore all be alive at one point: the LayerNorm backward that used to accumulate dx2 IN PLACE
        # into dy gets a copy to accumulate into (57 MB at cfg 3), the grouped launch sits before the last in-place accumulation.
        grouped = O.wgrad_group_pays(T, [(dy.shape[1], a.shape[1]), (f1.shape[1], h2.shape[1]), (dy.shape[1], o.shape[1]), (qkv.shape[1], h1.shape[1])])
        if f1.shape[1] % 8 == 0 and dy.shape[1] % 64 == 0:
            df1 = O.gemm_gelu_bwd(dy, W.w2, f1)  # quick-GELU backward in the fc2 dgrad's store phase (dy W2 never stored)
        else:
            df1 = O.quick_gelu_bwd(f1, O.gemm_nt(dy, W.w2, b_t=True))
        if not grouped:
            self._wgrad(dy, a, A.gview(p + "mlp.fc2.weight"), fresh, Tpad)
        O.colsum(dy, A.gview(p + "mlp.fc2.bias"), accumulate=acc)
        dh2 = O.gemm_nt(df1, W.w1, b_t=True)
        if not grouped:
            self._wgrad(df1, h2, A.gview(p + "mlp.fc1.weight"), fresh, Tpad)
        O.colsum(df1, A.gview(p + "mlp.fc1.bias"), accumulate=acc)
        dx2 = O.layernorm_bwd(x2, W.ln2w, dh2, eps, dx=O.copy2d(dy, torch.empty_like(dy)) if grouped else dy, accumulate_dx=True,
                              dw_out=A.gview(p + "layer_norm2.weight"), db_out=A.gview(p + "layer_norm2.bias"), accumulate=acc)
        do = O.gemm_nt(dx2, W.wo, b_t=True)
        if not grouped:
            self._wgrad(dx2, o, A.gview(p + "self_attn.out_proj.weight"), fresh, Tpad)
        O.colsum(dx2, A.gview(p + "self_attn.out_proj.bias"), accumulate=acc)
        dqkv = torch.empty_like(qkv)
        q, k, v = qkv[:, :vd], qkv[:, vd:2 * vd], qkv[:, 2 * vd:]
        O.attn_bwd2(q, k, v, o, do, lse, N, S, H, D, False, dq=dqkv[:, :vd], dk=dqkv[:, vd:2 * vd], dv=dqkv[:, 2 * vd:])
        dh1 = O.gemm_nt(dqkv, W.wqkv, b_t=True)
        gqkv = A.gspan(p + "self_attn.q_proj.weight", p + "self_attn.v_proj.weight", (3 * vd, vd))
        if grouped:
            O.wgrad_tn_grouped([(dy, a, A.gview(p + "mlp.fc2.weight")), (df1, h2, A.gview(p + "mlp.fc1.weight")),
                                (dx2, o, A.gview(p + "self_attn.out_proj.weight")), (dqkv, h1, gqkv)], accum=acc)
        else:
            self._wgrad(dqkv, h1, gqkv, fresh, Tpad)
        O.colsum(dqkv, A.gspan(p + "self_attn.q_proj.bias", p + "self_attn.v_proj.bias", (3 * vd,)), accumulate=acc)
        dx = O.layernorm_bwd(x, W.ln1w, dh1, eps, dx=dx2, accumulate_dx=True, dw_out=A.gview(p + "layer_norm1.weight"),
                             db_out=A.gview(p + "layer_norm1.bias"), accumulate=acc)
        self._ready(W.names)
        return dx

    # ---- the same encoder layer with its four Linears on the scaled-fp8 MFMA (fp8 training step, BASELINE cfg 5: "all Llama / ViT
    # Linear weights").  Operand formats as in the decoder (fp8_train_weights): activations / gradients row-quantised per token, the
    # transposed wgrad operands per feature (gradients) or tensor-wide (activations), weights per output channel + 128-block exponents.
    # LayerNorm, attention, quick-GELU, bias gradients and the residual stream stay 16-bit.
    def fp8_tower_weights(self, li):
        def make():
            out = []
            for W in self.vit:
                out.append(dict(wqkv=O.quant_fp8_rows_e4(W.wqkv), wo=O.quant_fp8_rows_e4(W.wo), w1=O.quant_fp8_rows_e4(W.w1), w2=O.quant_fp8_rows_e4(W.w2),
                                wqkvT=O.quant_fp8_rows_t_e4(W.wqkv), woT=O.quant_fp8_rows_t_e4(W.wo), w1T=O.quant_fp8_rows_t_e4(W.w1),
                                w2T=O.quant_fp8_rows_t_e4(W.w2)))
            return self._drop_zero_exponents(out)
        return self._derive("fp8_tower_weights", make)[li]

    def _vit_layer_fwd_fp8(self, W, li, x, N, S, vc, keep):
        H = vc.num_attention_heads
        vd = vc.hidden_size
        D = vd // H
        eps = vc.layer_norm_eps
        Q = self.fp8_tower_weights(li)
        dt = x.dtype
        h1 = O.layernorm_fwd(x, W.ln1w, W.ln1b, eps)
        a1 = O.quant_fp8_rows(h1)
        qkv = O.gemm_fp8(a1, Q["wqkv"], out_dtype=dt, bias=W.bqkv)
        q, k, v = qkv[:, :vd], qkv[:, vd:2 * vd], qkv[:, 2 * vd:]
        o, lse = O.attn_fwd2(q, k, v, N, S, H, D, causal=False)
        a2 = O.quant_fp8_rows(o)
        x2 = O.gemm_fp8(a2, Q["wo"], out_dtype=dt, bias=W.bo, resid=x)
        h2 = O.layernorm_fwd(x2, W.ln2w, W.ln2b, eps)
        a3 = O.quant_fp8_rows(h2)
        if keep:
            f1 = O.gemm_fp8(a3, Q["w1"], out_dtype=dt, bias=W.b1)
            a = O.quick_gelu_fwd(f1)
        else:
            f1 = None
            a = O.gemm_fp8(a3, Q["w1"], out_dtype=dt, bias=W.b1, act="quick_gelu")
        a4 = O.quant_fp8_rows(a)
        y = O.gemm_fp8(a4, Q["w2"], out_dtype=dt, bias=W.b2, resid=x2)
        return y, ((h1, qkv, o, lse, x2, h2, f1, a, (a1[1], a2[1], a3[1], a4[1])) if keep else None)

    def _vit_layer_bwd_fp8(self, W, li, x, dy, N, S, vc, saved, fresh):
        A = self.arena
        H = vc.num_attention_heads
        vd = vc.hidden_size
        D = vd // H
        eps = vc.layer_norm_eps
        if saved is None:
            _, saved = self._vit_layer_fwd_fp8(W, li, x, N, S, vc, keep=True)
        h1, qkv, o, lse, x2, h2, f1, a, (s_h1, s_o, s_h2, s_a) = saved
        Q = self.fp8_tower_weights(li)
        dt = x.dtype
        p = W.p
        acc = not fresh
        # fc2
        dy8, dyT8 = O.quant_fp8_both(dy)
        da = O.gemm_fp8(dy8, Q["w2T"], out_dtype=dt)
        self._wgrad_fp8(dyT8, a, s_a, A.gview(p + "mlp.fc2.weight"), fresh)
        O.colsum(dy, A.gview(p + "mlp.fc2.bias"), accumulate=acc)
        del dy8, dyT8
        df1 = O.quick_gelu_bwd(f1, da)
        df18, df1T8 = O.quant_fp8_both(df1)
        dh2 = O.gemm_fp8(df18, Q["w1T"], out_dtype=dt)
        self._wgrad_fp8(df1T8, h2, s_h2, A.gview(p + "mlp.fc1.weight"), fresh)
        O.colsum(df1, A.gview(p + "mlp.fc1.bias"), accumulate=acc)
        del df18, df1T8
        dx2 = O.layernorm_bwd(x2, W.ln2w, dh2, eps, dx=dy, accumulate_dx=True, dw_out=A.gview(p + "layer_norm2.weight"),
                              db_out=A.gview(p + "layer_norm2.bias"), accumulate=acc)
        dx28, dx2T8 = O.quant_fp8_both(dx2)
        do = O.gemm_fp8(dx28, Q["woT"], out_dtype=dt)
        self._wgrad_fp8(dx2T8, o, s_o, A.gview(p + "self_attn.out_proj.weight"), fresh)
        O.colsum(dx2, A.gview(p + "self_attn.out_proj.bias"), accumulate=acc)
        del dx28, dx2T8
        dqkv = torch.empty_like(qkv)
        q, k, v = qkv[:, :vd], qkv[:, vd:2 * vd], qkv[:, 2 * vd:]
        O.attn_bwd2(q, k, v, o, do, lse, N, S, H, D, False, dq=dqkv[:, :vd], dk=dqkv[:, vd:2 * vd], dv=dqkv[:, 2 * vd:])
        dqkv8, dqkvT8 = O.quant_fp8_both(dqkv)
        dh1 = O.gemm_fp8(dqkv8, Q["wqkvT"], out_dtype=dt)
        self._wgrad_fp8(dqkvT8, h1, s_h1, A.gspan(p + "self_attn.q_proj.weight", p + "self_attn.v_proj.weight", (3 * vd, vd)), fresh)
        O.colsum(dqkv, A.gspan(p + "self_attn.q_proj.bias", p + "self_attn.v_proj.bias", (3 * vd,)), accumulate=acc)
        dx = O.layernorm_bwd(x, W.ln1w, dh1, eps, dx=dx2, accumulate_dx=True, dw_out=A.gview(p + "layer_norm1.weight"),
                             db_out=A.gview(p + "layer_norm1.bias"), accumulate=acc)
        self._ready(W.names)
        return dx

    def tower(self, images, ctx=None):
        """list of [n_i,3,H,W] -> x [Nimg*(G2+1), vd] = hidden_states[select_layer] (CLS rows kept in place)."""
        m = self.model
        tower = m.get_model().vision_tower
        vc = tower.config
        A = self.arena
        dt = A.flat.dtype
        dev = A.flat.device
        G = vc.image_size // vc.patch_size
        G2 = G * G
        S = G2 + 1
        vd = vc.hidden_size
        K = 3 * vc.patch_size * vc.patch_size
        Kpad = _ru(K, 64)
        N = sum(int(im.shape[0]) for im in images)
        # patch rows are gathered straight into the tower's token-major layout [N, 1 + G2, Kpad] (zero CLS slot per image), one
        # launch per image tensor of the batch: no torch.cat of the pixels, and the weight gradient is one GEMM over all rows
        cols = torch.empty(N * S, Kpad, dtype=dt, device=dev)
        r = 0
        for im in images:
            im = im.to(device=dev)
            if im.dtype not in (torch.float32, dt):
                im = im.float()
            n = int(im.shape[0])
            O.im2col_patches(im.contiguous(), vc.patch_size, Kpad, dt, rows_per_img=S, row0=1, out=cols[r * S:(r + n) * S])
            r += n

        def _pad_patch_weight():  # K = 588 -> 640 (zero tail), once per weight version
            wp = torch.zeros(vd, Kpad, dtype=dt, device=dev)
            return O.copy2d(A.view(VT + "embeddings.patch_embedding.weight", shape=(vd, K)), wp)

        wpad = self._derive("patch_wpad", _pad_patch_weight)
        L = tower.layers_used
        train_tower = ctx is not None and ctx["train_tower"]
        xs, saves = [], []
        r32 = self.fp32_residual and not (ctx is not None and ctx.get("fp8"))
        cls_w, pos_w = A.view(VT + "embeddings.class_embedding"), A.view(VT + "embeddings.position_embedding.weight", shape=(S, vd))
        if r32:
            # the fp32 stream STARTS from fp32 values (round 4): patch projection stored in fp32, class / position embeddings added in fp32,
            # pre_layrnorm fp32 -> fp32; x0 = the 16-bit copy of its input the backward keeps.  (16-bit tensors at the start of the
            # stream were a sixth of the full-depth logits error: profiles/r04_parity_floor.txt)
            x0_32 = O.vit_assemble_f32(O.gemm_nt(cols, wpad, out_f32=True), cls_w, pos_w, N, G2)
            x32, x0 = O.layernorm_f32_to_f32(x0_32, A.view(VT + "pre_layrnorm.weight"), A.view(VT + "pre_layrnorm.bias"), vc.layer_norm_eps,
                                             want_x16=train_tower)
            del x0_32
            for i in range(L):
                x16, sv = self._vit_layer_fwd_r32(self.vit[i], x32, N, S, vc, keep=train_tower and self.save_activations, need_x16=train_tower)
                if train_tower:
                    xs.append(x16)
                saves.append(sv)
            x = O.convert(x32, torch.empty(x32.shape, dtype=dt, device=dev))  # hidden_states[select_layer] as the projector's 16-bit GEMM operand
            del x32
            L = 0
        else:
            x0 = O.vit_assemble(O.gemm_nt(cols, wpad), cls_w, pos_w, N, G2)
            x = O.layernorm_fwd(x0, A.view(VT + "pre_layrnorm.weight"), A.view(VT + "pre_layrnorm.bias"), vc.layer_norm_eps)
        fp8_tower = bool(ctx is not None and ctx.get("fp8_train") and self.fp8_tower and not r32 and vd % 128 == 0 and
                         vc.intermediate_size % 128 == 0)
        for i in range(L):
            if train_tower:
                xs.append(x)
            if fp8_tower:
                x, sv = self._vit_layer_fwd_fp8(self.vit[i], i, x, N, S, vc, keep=train_tower and self.save_activations)
            else:
                x, sv = self._vit_layer_fwd(self.vit[i], x, N, S, vc, keep=train_tower and self.save_activations)
            saves.append(sv)
        if ctx is not None:
            ctx["fp8_tower"] = fp8_tower
        if ctx is not None:
            ctx.update(vit_cols=cols if train_tower else None, vit_x0=x0 if train_tower else None, vit_xs=xs, vit_saves=saves,
                       vit_N=N, vit_S=S, vit_Kpad=Kpad)
        return x, N, S

    def tower_bwd(self, ctx, dx, fresh):
        m = self.model
        tower = m.get_model().vision_tower
        vc = tower.config
        A = self.arena
        N, S = ctx["vit_N"], ctx["vit_S"]
        L = tower.layers_used
        for i in reversed(range(L)):
            if ctx.get("fp8_tower"):
                dx = self._vit_layer_bwd_fp8(self.vit[i], i, ctx["vit_xs"][i], dx, N, S, vc, ctx["vit_saves"][i], fresh)
            else:
                dx = self._vit_layer_bwd(self.vit[i], ctx["vit_xs"][i], dx, N, S, vc, ctx["vit_saves"][i], fresh)
            ctx["vit_xs"][i] = None
            ctx["vit_saves"][i] = None
        acc = not fresh
        vd = vc.hidden_size
        G2 = S - 1
        dx0 = O.layernorm_bwd(ctx["vit_x0"], A.view(VT + "pre_layrnorm.weight"), dx, vc.layer_norm_eps,
                              dw_out=A.gview(VT + "pre_layrnorm.weight"), db_out=A.gview(VT + "pre_layrnorm.bias"), accumulate=acc)
        d3 = dx0.view(N, S, vd)
        # position embedding / class embedding grads: sums over images (column sums of [N, S*vd] and of the CLS rows)
        O.colsum(dx0.view(N, S * vd), A.gview(VT + "embeddings.position_embedding.weight").view(S * vd), accumulate=acc)
        O.colsum(d3[:, 0, :], A.gview(VT + "embeddings.class_embedding"), accumulate=acc)
        # patch embedding weight grad: dW[vd, Kpad] = dx0^T @ cols over ALL token rows (the CLS rows of cols are zero): one
        # K-strided MFMA GEMM on the operands as they lie in memory, then the K-padding is dropped by a strided copy
        K = 3 * vc.patch_size * vc.patch_size
        Kpad = ctx["vit_Kpad"]
        gw = torch.empty(vd, Kpad, dtype=dx0.dtype, device=dx0.device)
        O.wgrad_tn(dx0, ctx["vit_cols"], gw, accum=False)
        O.copy2d(gw[:, :K], A.gview(VT + "embeddings.patch_embedding.weight").view(vd, K), accumulate=acc)
        self._ready([n for n in A.names if n.startswith(VT + "embeddings.") or n.startswith(VT + "pre_layrnorm")])

    # ------------------------------------------------------------------------------------------
    # projector
    # ------------------------------------------------------------------------------------------
    def projector(self, x, N, S, ctx=None, out_f32=False):
        """x [N*S, vd] (tower output incl. CLS rows) -> (feats2d, rows_per_img, row0, P).  out_f32: the features stay fp32 (they are spliced
        straight into the decoder's fp32 residual stream)."""
        m = self.model
        proj = m.get_model().projector
        A = self.arena
        kind = "conv" if hasattr(proj, "conv_stride") else "mlp"
        tower = m.get_model().vision_tower
        cls_keep = tower.select_feature == "cls_patch"
        if kind == "mlp":
            feats = O.gemm_nt(x, A.view("model.projector.projector.weight"), bias=A.view("model.projector.projector.bias"), out_f32=out_f32)
            if ctx is not None:
                ctx.update(proj_in=x)
            return feats, S, (0 if cls_keep else 1), (S if cls_keep else S - 1)
        # ConvProjector (conv_projector.py:23-39): Conv2d(vd -> d, k3, stride, pad 1) over the G x G patch grid as an
        # implicit GEMM: gather -> MFMA GEMM with weight.view(d, vd*9) in place (+bias) -> [N*(G/s)^2, d]
        if cls_keep:
            raise NotImplementedError("ConvProjector needs the square patch grid (vision_select_feature='patch')")
        vd = x.shape[1]
        G = int(round(math.sqrt(S - 1)))
        stride = proj.conv_stride
        cols = O.conv3x3_cols(x, N, G, vd, stride, S, 1)
        w = A.view("model.projector.projector.weight", shape=(A.params["model.projector.projector.weight"].shape[0], vd * 9))
        feats = O.gemm_nt(cols, w, bias=A.view("model.projector.projector.bias"), out_f32=out_f32)
        Go = (G + 2 - 3) // stride + 1
        if ctx is not None:
            ctx.update(proj_in=cols, proj_conv=(N, G, vd, stride, S))
        return feats, Go * Go, 0, Go * Go

    def projector_bwd(self, ctx, dfeats, fresh):
        """returns dx for the tower output ([N*S, vd])."""
        A = self.arena
        x = ctx["proj_in"]
        wname, bname = "model.projector.projector.weight", "model.projector.projector.bias"
        dx = None
        if "proj_conv" in ctx:
            N, G, vd, stride, S = ctx["proj_conv"]
            w = A.view(wname, shape=(A.params[wname].shape[0], vd * 9))
            if ctx["train_tower"]:
                dcols = O.gemm_nt(dfeats, w, b_t=True)  # [rows, vd*9]
                dx = O.conv3x3_col2im(dcols, N, G, vd, stride, S, 1)
            if self._trainable(wname):
                self._wgrad(dfeats, x, A.gview(wname).view(w.shape), fresh, _ru(x.shape[0], 64))
                O.colsum(dfeats, A.gview(bname), accumulate=not fresh)
                self._ready([wname, bname])
            return dx
        if ctx["train_tower"]:
            dx = O.gemm_nt(dfeats, A.view(wname), b_t=True)
        if self._trainable(wname):
            T = x.shape[0]
            self._wgrad(dfeats, x, A.gview(wname), fresh, _ru(T, 64))
            O.colsum(dfeats, A.gview(bname), accumulate=not fresh)
            self._ready([wname, bname])
        return dx

    # ------------------------------------------------------------------------------------------
    # Llama
    # ------------------------------------------------------------------------------------------
    def _llama_layer_fwd(self, W, x, B, S, lens, keep, kv_out=None, unpad=None):
        cfg = self.model.config
        d, H, D = cfg.hidden_size, cfg.num_attention_heads, head_dim_of(cfg)
        eps = cfg.rms_norm_eps
        h1 = O.rmsnorm_fwd(x, W.ln1, eps)
        qkv = O.gemm_nt_rope(h1, W.wqkv, self.rope, S, H, D)  # q|k|v projection with RoPE in the GEMM epilogue
        o, lse, packed = self._attn_fwd(qkv, B, S, H, D, lens, unpad, kv_out)
        x2 = O.gemm_nt(o, W.wo, resid=x)
        h2 = O.rmsnorm_fwd(x2, W.ln2, eps)
        gu, act = O.gemm_swiglu_fwd(h2, W.wgu)  # gate|up projection; SwiGLU in the same launch's epilogue
        y = O.gemm_nt(act, W.wd, resid=x2)
        return y, (self._slim(W, (h1, qkv, o, lse, x2, h2, gu, act, packed)) if keep else None)

    def _slim(self, W, saved):
        """mem_level: drop what the backward can re-derive (see __init__)."""
        if not self.mem_level:
            return saved
        h1, qkv, o, lse, x2, h2, gu, act, packed = saved
        drop_act = self.mem_level >= 2 and int(W.p.split(".")[2]) < self.mem_act_layers
        return (None, qkv, o, lse, x2, None, gu, None if drop_act else act, packed)

    def _llama_layer_fwd_r32(self, W, x32, B, S, lens, keep, kv_out=None, unpad=None, need_x16=True):
        """_llama_layer_fwd on the fp32 residual stream: x32 [T, d] is updated IN PLACE; returns (x16 = the 16-bit copy of the layer
        input for the backward, saved activations | None)."""
        cfg = self.model.config
        H, D = cfg.num_attention_heads, head_dim_of(cfg)
        eps = cfg.rms_norm_eps
        h1, x16 = O.norm_fwd_f32in(x32, W.ln1, eps, want_x16=need_x16)  # (the 16-bit copy of the layer input is the BACKWARD's: not written without one)
        qkv = O.gemm_nt_rope(h1, W.wqkv, self.rope, S, H, D)
        o, lse, packed = self._attn_fwd(qkv, B, S, H, D, lens, unpad, kv_out)
        O.gemm_nt(o, W.wo, out=x32, accum=True)           # x += o Wo^T, fp32 read-modify-write in the GEMM epilogue
        h2, x2_16 = O.norm_fwd_f32in(x32, W.ln2, eps, want_x16=keep)
        gu, act = O.gemm_swiglu_fwd(h2, W.wgu)
        O.gemm_nt(act, W.wd, out=x32, accum=True)         # x += act Wd^T
        return x16, (self._slim(W, (h1, qkv, o, lse, x2_16, h2, gu, act, packed)) if keep else None)

    def _vit_layer_fwd_r32(self, W, x32, N, S, vc, keep, need_x16=True):
        H = vc.num_attention_heads
        vd = vc.hidden_size
        D = vd // H
        eps = vc.layer_norm_eps
        h1, x16 = O.norm_fwd_f32in(x32, W.ln1w, eps, b=W.ln1b, want_x16=need_x16)
        qkv = O.gemm_nt(h1, W.wqkv, bias=W.bqkv)
        q, k, v = qkv[:, :vd], qkv[:, vd:2 * vd], qkv[:, 2 * vd:]
        o, lse = O.attn_fwd2(q, k, v, N, S, H, D, causal=False)
        O.gemm_nt(o, W.wo, bias=W.bo, out=x32, accum=True)
        h2, x2_16 = O.norm_fwd_f32in(x32, W.ln2w, eps, b=W.ln2b, want_x16=keep)
        if keep:
            f1, a = self._fc1_gelu(h2, W)
        else:
            f1 = None
            a = O.gemm_nt(h2, W.w1, bias=W.b1, act="quick_gelu")
        O.gemm_nt(a, W.w2, bias=W.b2, out=x32, accum=True)
        return x16, (((None if self.mem_level else h1), qkv, o, lse, x2_16, (None if self.mem_level else h2), f1, a) if keep else None)

    # ---- attention under a key-padding mask (llama_flash_attn_monkey_patch.py:87-102) ---------------------------------------------
    # Right-padded batches (the collator's, collator.py:29-34) only need per-sample lengths: the kernels skip keys >= lens[b] and
    # zero the padded query rows.  Any other mask takes the reference's own route: unpad_input (gather the valid rows of the ROTATED
    # q|k|v - positions stay absolute, RoPE ran in the projection's epilogue) -> causal varlen attention over the packed tokens ->
    # pad_input (scatter back, zeros elsewhere), with the same two gathers around the backward.
    def _attn_fwd(self, qkv, B, S, H, D, lens, unpad, kv_out=None):
        d = H * D
        if unpad is None:
            q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
            if kv_out is not None:  # prefill: rotated keys and values go to the decode cache [B, Smax, d]
                kv_out[0][:, :S].copy_(k.view(B, S, d))
                kv_out[1][:, :S].copy_(v.view(B, S, d))
            o, lse = O.attn_fwd2(q, k, v, B, S, H, D, causal=True, seqlens=lens)
            return o, lse, None
        fwd, inv, cnt = unpad
        qkv_c = O.gather_rows2d(qkv, fwd, torch.empty_like(qkv))
        q, k, v = qkv_c[:, :d], qkv_c[:, d:2 * d], qkv_c[:, 2 * d:]
        if kv_out is not None:  # the cache keeps only the VALID keys (rows 0..count-1); their rotation already carries the position
            kv_out[0][:, :S].copy_(k.view(B, S, d))
            kv_out[1][:, :S].copy_(v.view(B, S, d))
        o_c, lse = O.attn_fwd2(q, k, v, B, S, H, D, causal=True, seqlens=cnt)
        o = O.gather_rows2d(o_c, inv, torch.empty_like(o_c))
        return o, lse, (qkv_c, o_c)

    def _attn_bwd(self, qkv, o, do, lse, B, S, H, D, lens, unpad, packed):
        """-> dqkv [T, 3 H D] w.r.t. the UN-rotated q, k (inverse RoPE applied)."""
        d = H * D
        dqkv = torch.empty_like(qkv)
        if unpad is None:
            q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
            O.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, True, seqlens=lens, dq=dqkv[:, :d], dk=dqkv[:, d:2 * d], dv=dqkv[:, 2 * d:],
                        rope=self.rope)  # inverse RoPE of dq, dk fused into the kernels' epilogues
            return dqkv
        fwd, inv, cnt = unpad
        if packed is None:  # (layer recompute dropped them)
            qkv_c = O.gather_rows2d(qkv, fwd, torch.empty_like(qkv))
            o_c = O.gather_rows2d(o, fwd, torch.empty_like(o))
        else:
            qkv_c, o_c = packed
        do_c = O.gather_rows2d(do, fwd, torch.empty_like(do))
        q, k, v = qkv_c[:, :d], qkv_c[:, d:2 * d], qkv_c[:, 2 * d:]
        dqkv_c = torch.empty_like(qkv)
        O.attn_bwd2(q, k, v, o_c, do_c, lse, B, S, H, D, True, seqlens=cnt, dq=dqkv_c[:, :d], dk=dqkv_c[:, d:2 * d], dv=dqkv_c[:, 2 * d:])
        O.gather_rows2d(dqkv_c, inv, dqkv)
        return O.rope_qk_(dqkv, self.rope, S, H, D, inverse=True)  # positions are those of the UNPACKED rows

    def quantize_forward_weights(self):
        """fp8 (e4m3, one scale per output channel) copies of the decoder's Linear weights for the fp8 FORWARD (inference /
        prefill form of BASELINE cfg 5's fp8 MFMA weight path).  Re-run after the weights change."""
        self.ensure_arena()
        self._fp8_fwd = self._drop_zero_exponents([dict(wqkv=O.quant_fp8_rows_e4(W.wqkv), wo=O.quant_fp8_rows_e4(W.wo), wgu=O.quant_fp8_rows_e4(W.wgu),
                                                        wd=O.quant_fp8_rows_e4(W.wd)) for W in self.llama])
        return self._fp8_fwd

    def _llama_layer_fwd_fp8(self, W, Q, x, B, S, lens, kv_out=None, unpad=None):
        """Decoder layer with every Linear on the scaled-fp8 MFMA: activations are quantised per token row right before
        each GEMM (dynamic scaling), weights per output channel (once); residual stream, norms, RoPE, attention and
        SwiGLU stay 16-bit.  Forward only."""
        cfg = self.model.config
        d, H, D = cfg.hidden_size, cfg.num_attention_heads, head_dim_of(cfg)
        eps = cfg.rms_norm_eps
        _, a1 = O.rmsnorm_fwd_q8(x, W.ln1, eps)
        qkv = O.gemm_fp8_rope(a1, Q["wqkv"], self.rope, S, H, D, out_dtype=x.dtype)
        o, _, _ = self._attn_fwd(qkv, B, S, H, D, lens, unpad, kv_out)
        x2 = O.gemm_fp8(O.quant_fp8_rows(o), Q["wo"], out_dtype=x.dtype, resid=x)
        _, a3 = O.rmsnorm_fwd_q8(x2, W.ln2, eps)
        _, act = O.gemm_fp8_swiglu_fwd(a3, Q["wgu"], out_dtype=x.dtype)
        return O.gemm_fp8(O.quant_fp8_rows(act), Q["wd"], out_dtype=x.dtype, resid=x2)

    # ---- fp8 TRAINING step (BASELINE cfg 5: "fp8 MFMA weight path"; no reference counterpart) -------------
    # Every decoder Linear - forward, dgrad and wgrad - runs on the scaled-fp8 MFMA as an NT product of two ROW-quantised
    # e4m3 operands (include/merlin_hip.h, "fp8 TRAINING step"): activations / gradients are quantised dynamically per row
    # (per token for forward and dgrad, per feature - on the transposed copy - for wgrad), weights per output channel (W8)
    # and per input channel (WT8 = rowquant(W^T)), re-quantised once per weight version.  Residual stream, norms, RoPE,
    # attention, SwiGLU, CE and all gradient accumulators stay 16-bit / fp32 as in the bf16 step; lm_head (fp8_head_weights) and the
    # CLIP tower's Linears (_vit_layer_fwd_fp8) join the decoder's on the fp8 MFMA.
    def fp8_train_weights(self, li):
        def make():
            out = []
            for W in self.llama:
                # weights: per-row fp32 scale + per-128-block exponents (applied by the MFMA's block-scale operand)
                out.append(dict(wqkv=O.quant_fp8_rows_e4(W.wqkv), wo=O.quant_fp8_rows_e4(W.wo), wgu=O.quant_fp8_rows_e4(W.wgu), wd=O.quant_fp8_rows_e4(W.wd),
                                wqkvT=O.quant_fp8_rows_t_e4(W.wqkv), woT=O.quant_fp8_rows_t_e4(W.wo), wguT=O.quant_fp8_rows_t_e4(W.wgu),
                                wdT=O.quant_fp8_rows_t_e4(W.wd)))
            return self._drop_zero_exponents(out)
        return self._derive("fp8_train_weights", make)[li]

    def fp8_head_weights(self):
        """lm_head.weight [V, d] for the fp8 training step: per-output-channel copy (logits) and the copy of its transpose (dgrad)."""
        def make():
            cfg = self.model.config
            V, d = cfg.vocab_size, cfg.hidden_size
            wlm = self.arena.view("lm_head.weight", numel=_ru(V, 64) * d, shape=(_ru(V, 64), d))  # incl. the zero pad rows
            # (the transposed copy is quantised per input channel only: its rows are the whole vocabulary long, beyond what the
            #  block-exponent image of the 8-wave kernel holds)
            return self._drop_zero_exponents([dict(w=O.quant_fp8_rows_e4(wlm), wT=O.quant_fp8_rows_t(wlm))])[0]
        return self._derive("fp8_head_weights", make)

    @staticmethod
    def _drop_zero_exponents(layers):
        """The quantiser leaves a device flag per weight: "some 128-block exponent is non-zero".  Read all of them back ONCE per weight
        version (one host sync per optimizer step) and drop the exponent image of every weight whose blocks all sit within a factor 2
        of their row maximum (i.i.d.-like weights: all of them) - such operands may take the 4-wave fp8 GEMM (csrc/gemm_w4.hip), which
        applies per-row scales only; the others keep the image and the 8-wave kernel's block-scaled loop."""
        trip = [(d, k) for d in layers for k, v in d.items() if len(v) > 2 and v[2] is not None]
        if not trip:
            return layers
        flags = torch.stack([d[k][2][:4].view(torch.int32) for d, k in trip]).view(-1).cpu()
        for (d, k), f in zip(trip, flags.tolist()):
            if f == 0:
                d[k] = d[k][:2]
        return layers

    def _llama_layer_fwd_fp8_train(self, W, li, x, B, S, lens, keep, unpad=None):
        cfg = self.model.config
        d, H, D = cfg.hidden_size, cfg.num_attention_heads, head_dim_of(cfg)
        eps = cfg.rms_norm_eps
        Q = self.fp8_train_weights(li)
        h1, a1 = O.rmsnorm_fwd_q8(x, W.ln1, eps)  # norm + row quantisation of its output in one launch
        qkv = O.gemm_fp8_rope(a1, Q["wqkv"], self.rope, S, H, D, out_dtype=x.dtype)
        o, lse, packed = self._attn_fwd(qkv, B, S, H, D, lens, unpad)
        a2 = O.quant_fp8_rows(o)
        x2 = O.gemm_fp8(a2, Q["wo"], out_dtype=x.dtype, resid=x)
        h2, a3 = O.rmsnorm_fwd_q8(x2, W.ln2, eps)
        gu, act = O.gemm_fp8_swiglu_fwd(a3, Q["wgu"], out_dtype=x.dtype)
        a4 = O.quant_fp8_rows(act)
        y = O.gemm_fp8(a4, Q["wd"], out_dtype=x.dtype, resid=x2)
        # the row scales (one float per token) are kept: their maximum is the tensor-wide scale of the transposed wgrad operand
        return y, ((h1, qkv, o, lse, x2, h2, gu, act, (a1[1], a2[1], a3[1], a4[1]), packed) if keep else None)

    def _wgrad_fp8(self, dyT8, x, sx, gout, fresh):
        """gout[N_out, K_in] (+)= dy^T x on the scaled-fp8 MFMA: both operands as transposed e4m3 copies (contraction over the
        tokens, zero-padded to a multiple of 128).  The GRADIENT operand is scaled per output feature (column maxima: a pass
        over dy - the q / k / v thirds of dqkv differ by orders of magnitude, a tensor-wide scale flushes the small ones to
        zero, measured - shared with the row quantisation of the same tensor for the dgrad: O.quant_fp8_both reads it twice in
        all); the ACTIVATION operand (norm outputs, attention output, SwiGLU output: homogeneous columns) by its
        tensor-wide scale = the largest of the row scales its forward quantisation already produced (single pass)."""
        O.gemm_fp8(dyT8, O.quant_fp8_t_from_rows(x, sx), out=gout, accum=not fresh)

    def _llama_layer_bwd_fp8(self, W, li, x, dy, B, S, lens, saved, fresh, unpad=None):
        cfg = self.model.config
        A = self.arena
        d, ff, H, D = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, head_dim_of(cfg)
        eps = cfg.rms_norm_eps
        if saved is None:
            _, saved = self._llama_layer_fwd_fp8_train(W, li, x, B, S, lens, keep=True, unpad=unpad)
        h1, qkv, o, lse, x2, h2, gu, act, (s_h1, s_o, s_h2, s_act), packed = saved
        Q = self.fp8_train_weights(li)
        p = W.p
        acc = not fresh
        train = self._trainable(p + "mlp.down_proj.weight")
        dt = x.dtype
        dy8, dyT8 = O.quant_fp8_both(dy) if train else (O.quant_fp8_rows(dy), None)
        # SwiGLU backward in the dgrad's store phase - which also takes the row / column maxima its quantiser needs (one read of dgu less)
        dgu, dgu_amax = O.gemm_fp8_swiglu_bwd(dy8, Q["wdT"], gu, want_amax=True) if train and self.fp8_fused_amax else (O.gemm_fp8_swiglu_bwd(dy8, Q["wdT"], gu), None)
        if train:
            self._wgrad_fp8(dyT8, act, s_act, A.gview(p + "mlp.down_proj.weight"), fresh)
        del act, gu, dy8, dyT8
        dgu8, dguT8 = O.quant_fp8_both(dgu, amax=dgu_amax) if train else (O.quant_fp8_rows(dgu), None)
        dh2 = O.gemm_fp8(dgu8, Q["wguT"], out_dtype=dt)
        if train:
            self._wgrad_fp8(dguT8, h2, s_h2, A.gspan(p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight", (2 * ff, d)), fresh)
        del dgu, dgu8, dguT8
        dx2 = O.rmsnorm_bwd(x2, W.ln2, dh2, eps, dx=dy, accumulate_dx=True,
                            dw_out=A.gview(p + "post_attention_layernorm.weight") if train else None, dw_accumulate=acc)
        dx28, dx2T8 = O.quant_fp8_both(dx2) if train else (O.quant_fp8_rows(dx2), None)
        do = O.gemm_fp8(dx28, Q["woT"], out_dtype=dt)
        if train:
            self._wgrad_fp8(dx2T8, o, s_o, A.gview(p + "self_attn.o_proj.weight"), fresh)
        del dx28, dx2T8
        dqkv = self._attn_bwd(qkv, o, do, lse, B, S, H, D, lens, unpad, packed)
        dqkv8, dqkvT8 = O.quant_fp8_both(dqkv) if train else (O.quant_fp8_rows(dqkv), None)
        dh1 = O.gemm_fp8(dqkv8, Q["wqkvT"], out_dtype=dt)
        if train:
            self._wgrad_fp8(dqkvT8, h1, s_h1, A.gspan(p + "self_attn.q_proj.weight", p + "self_attn.v_proj.weight", (3 * d, d)), fresh)
        dx = O.rmsnorm_bwd(x, W.ln1, dh1, eps, dx=dx2, accumulate_dx=True,
                           dw_out=A.gview(p + "input_layernorm.weight") if train else None, dw_accumulate=acc)
        if train:
            self._ready(W.names)
        return dx

    def _llama_layer_bwd(self, W, x, dy, B, S, lens, saved, fresh, unpad=None, rows=None):
        cfg = self.model.config
        A = self.arena
        d, ff, H, D = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, head_dim_of(cfg)
        eps = cfg.rms_norm_eps
        if saved is None:
            _, saved = self._llama_layer_fwd(W, x, B, S, lens, keep=True, unpad=unpad)
        h1, qkv, o, lse, x2, h2, gu, act, packed = saved
        T = x.shape[0]
        Tpad = _ru(T, 64)
        p = W.p
        acc = not fresh
        train = self._trainable(p + "mlp.down_proj.weight")
        if rows is not None:
            # LAST decoder layer under a sparse loss (engine.sparse_head): dy - the gradient of this layer's output - is exactly zero on every
            # row the loss does not score (the head's dgrad gathered zero rows there and RMSNorm backward is row-wise), so the MLP half, the
            # post-attention norm and the o projection of this layer contract / map over the scored rows alone: the rows are gathered into
            # compact [npad, .] operands (rows_f: flat positions, -1 = zero pad row), run through the same kernels, and the two results the
            # rest of the backward needs dense - d(attention output) and the residual gradient - are gathered back (rows_i; zero rows elsewhere).
            rows_f, rows_i = rows
            npad = rows_f.numel()
            g2 = lambda t: O.gather_rows2d(t, rows_f, torch.empty(npad, t.shape[1], dtype=t.dtype, device=t.device))  # noqa: E731
            dy_c, gu_c, x2_c = g2(dy), g2(gu), g2(x2)
            dgu_c = O.gemm_swiglu_bwd(dy_c, W.wd, gu_c)
            if train:
                act_c = g2(act) if act is not None else O.swiglu_fwd(gu_c)
                self._wgrad(dy_c, act_c, A.gview(p + "mlp.down_proj.weight"), fresh, npad)
                del act_c
            del act, gu, gu_c
            dh2_c = O.gemm_nt(dgu_c, W.wgu, b_t=True)
            if train:
                h2_c = g2(h2) if h2 is not None else O.rmsnorm_fwd(x2_c, W.ln2, eps)
                self._wgrad(dgu_c, h2_c, A.gspan(p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight", (2 * ff, d)), fresh, npad)
                del h2_c
            del dgu_c, h2
            dx2_c = O.rmsnorm_bwd(x2_c, W.ln2, dh2_c, eps, dx=dy_c, accumulate_dx=True,
                                  dw_out=A.gview(p + "post_attention_layernorm.weight") if train else None, dw_accumulate=acc)
            do_c = O.gemm_nt(dx2_c, W.wo, b_t=True)
            if train:
                self._wgrad(dx2_c, g2(o), A.gview(p + "self_attn.o_proj.weight"), fresh, npad)
            do = O.gather_rows2d(do_c, rows_i, torch.empty(T, d, dtype=do_c.dtype, device=do_c.device))
            dx2 = O.gather_rows2d(dx2_c, rows_i, dy)  # (dy's storage: its values live on in dx2_c)
            del dy_c, x2_c, dh2_c, dx2_c, do_c
        else:
            dgu = O.gemm_swiglu_bwd(dy, W.wd, gu)  # dact = dy Wd never leaves the kernel: SwiGLU backward in the epilogue
            if train:
                if act is None:  # mem_level 2: not kept
                    act = O.swiglu_fwd(gu)
                self._wgrad(dy, act, A.gview(p + "mlp.down_proj.weight"), fresh, Tpad)
            del act
            dh2 = O.gemm_nt(dgu, W.wgu, b_t=True)
            if train:
                if h2 is None:  # mem_level >= 1: the normed operand is re-derived from the saved 16-bit layer-half input
                    h2 = O.rmsnorm_fwd(x2, W.ln2, eps)
                self._wgrad(dgu, h2, A.gspan(p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight", (2 * ff, d)), fresh, Tpad)
            del dgu, gu, h2
            dx2 = O.rmsnorm_bwd(x2, W.ln2, dh2, eps, dx=dy, accumulate_dx=True,
                                dw_out=A.gview(p + "post_attention_layernorm.weight") if train else None, dw_accumulate=acc)
            do = O.gemm_nt(dx2, W.wo, b_t=True)
            if train:
                self._wgrad(dx2, o, A.gview(p + "self_attn.o_proj.weight"), fresh, Tpad)
        dqkv = self._attn_bwd(qkv, o, do, lse, B, S, H, D, lens, unpad, packed)
        dh1 = O.gemm_nt(dqkv, W.wqkv, b_t=True)
        if train:
            if h1 is None:
                h1 = O.rmsnorm_fwd(x, W.ln1, eps)
            self._wgrad(dqkv, h1, A.gspan(p + "self_attn.q_proj.weight", p + "self_attn.v_proj.weight", (3 * d, d)), fresh, Tpad)
        del h1
        dx = O.rmsnorm_bwd(x, W.ln1, dh1, eps, dx=dx2, accumulate_dx=True,
                           dw_out=A.gview(p + "input_layernorm.weight") if train else None, dw_accumulate=acc)
        if train:
            self._ready(W.names)
        return dx

    # ------------------------------------------------------------------------------------------
    # splice / head
    # ------------------------------------------------------------------------------------------
    def _check_errors(self):
        """Reads the device-side validation flags back (raises like the reference would) and returns True when the attention
        mask is NOT a right-padded prefix (left padding, holes): the caller then takes the unpad / pad attention path."""
        self._mask_all_ones = False
        if self._err is None:
            return False
        err, ev = self._err
        self._err = None
        ev.synchronize()
        e = err.tolist()
        if e[0]:
            raise ValueError(f"The number of image start tokens and image end tokens should be the same (sample {e[2]}, difference {e[3]}).")
        if e[1]:
            raise ValueError(f"The image end token should follow the image start token (sample {e[2]}, <im_start> at {e[3]}).")
        if e[4]:
            raise IndexError(f"index out of range in self: input_ids holds an id outside [0, vocab_size) at flat position {e[5]}")
        if e[6]:
            raise IndexError(f"Target out of bounds: labels holds a value that is neither -100 nor in [0, vocab_size) at flat position {e[7]}")
        self._mask_all_ones = not e[10]
        return bool(e[8])

    def _splice_geometry(self):
        """(rows per image in the projector output, first patch row, P = image tokens per image) from the configs alone, so the
        index / validation kernels can run BEFORE the tower (their flags are read back while the GPU is busy with it)."""
        inner = self.model.get_model()
        tower, proj = inner.vision_tower, inner.projector
        vc = tower.config
        G = vc.image_size // vc.patch_size
        if hasattr(proj, "conv_stride"):
            Go = (G + 2 - 3) // proj.conv_stride + 1
            return Go * Go, 0, Go * Go
        S = G * G + 1
        cls_keep = tower.select_feature == "cls_patch"
        return S, (0 if cls_keep else 1), (S if cls_keep else S - 1)

    def validate_and_index(self, input_ids, labels, mask, lens, images, use_images):
        """One pass of device-side input checks (+ the splice row table when the batch carries images); returns src | None."""
        m = self.model
        dev = self.arena.flat.device
        err_dev = torch.zeros(12, dtype=torch.int32, device=dev)
        src = None
        if use_images:
            rpi, row0, P = self._splice_geometry()
            counts = [int(im.shape[0]) for im in images]
            off = [0]
            for c in counts:
                off.append(off[-1] + c)
            B = input_ids.shape[0]
            off = (off + [off[-1]] * (B + 1))[: B + 1]  # fewer image entries than samples: zip() semantics
            img_off = torch.tensor(off, dtype=torch.int32).to(dev, non_blocking=True)
            src = O.splice_index(input_ids, img_off, P, m.im_patch_token, m.im_start_token, m.im_end_token, err_dev,
                                 rows_per_img=rpi, row0=row0)
        # strict_checks = False drops the id / label range checks, never the mask classification: whether the mask is a right-padded
        # prefix (lens fast path) or not (unpad / pad path) decides what the attention kernels compute, and generate() follows the
        # same decision (prefill -> cache.rpos), so it is always made here, on the device, from the mask itself
        if self.strict_checks or mask is not None:
            if self.strict_checks and (input_ids is not None or labels is not None or mask is not None):
                O.check_inputs(input_ids, labels, mask, lens, err_dev, m.config.vocab_size)
            elif mask is not None:
                O.check_inputs(None, None, mask, lens, err_dev, m.config.vocab_size)
            if getattr(self, "_err_host", None) is None:
                self._err_host = torch.empty(12, dtype=torch.int32, pin_memory=True)  # reused: every forward consumes its own check
            host = self._err_host
            host.copy_(err_dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._err = (host, ev)
        return src

    # ------------------------------------------------------------------------------------------
    # public entry points
    # ------------------------------------------------------------------------------------------
    def forward(self, input_ids, attention_mask, labels, images, inputs_embeds=None, want_grad=False, loss_only=False,
                kv_cache=None, last_only=False, fp8=False):
        """Returns (loss fp32 scalar tensor | None, logits fp32 [B,S,V] view | None, ctx)."""
        m = self.model
        cfg = m.config
        A = self.ensure_arena()
        dt, dev = A.flat.dtype, A.flat.device
        inner = m.get_model()
        tower = getattr(inner, "vision_tower", None)
        B, S = (input_ids.shape if input_ids is not None else inputs_embeds.shape[:2])
        T = B * S
        d = cfg.hidden_size
        ctx = {"B": B, "S": S, "want_grad": want_grad, "fp8_train": fp8 == "train", "fp8": bool(fp8)}
        ctx["train_tower"] = bool(want_grad and tower is not None and not tower.freeze_vision_tower and
                                  any(p.requires_grad for p in tower.parameters()))
        self._rope_table(S, dev)
        if input_ids is not None:
            input_ids = input_ids.to(dev).contiguous()
        lens = None
        if attention_mask is not None:
            am = attention_mask.to(dev)
            if am.dtype != torch.bool:
                am = am != 0
            lens = O.mask_lens(am.contiguous())
        ctx["lens"] = lens
        if labels is not None:
            labels = labels.to(dev).contiguous()
        # ---- multimodal splice (base_mmgpt.py:82-165) ----
        feats = None
        use_images = tower is not None and images is not None and input_ids is not None and S != 1
        src = self.validate_and_index(input_ids, labels, am.contiguous() if attention_mask is not None else None, lens, images, use_images)
        r32 = self.fp32_residual and not fp8  # (the fp8 paths run on 16-bit streams)
        if use_images:
            xt, N, Sv = self.tower(images, ctx)
            feats, rpi, row0, P = self.projector(xt, N, Sv, ctx, out_f32=r32 and inputs_embeds is None)
            assert (rpi, row0, P) == self._splice_geometry()
            ctx.update(src=src, n_feat_rows=feats.shape[0])
        general_mask = self._check_errors()  # the flags were produced before the tower was enqueued: no wait for compute
        unpad = None
        if attention_mask is not None and (general_mask or self.force_unpad):
            unpad = O.mask_unpad_index(am.contiguous())
        elif attention_mask is not None and self._mask_all_ones and not self.keep_full_lengths:
            # the collator's mask of a batch without padding (cfg 3 / cfg 5: every sequence fills the context) is all ones: drop the lengths, so the
            # attention kernels take their no-lengths forms (llama_flash_attn_monkey_patch.py:76-85, the `key_padding_mask is None` branch:
            # cu_seqlens = arange) - the flag comes from the same device-side check as the mask classification, read back without a wait
            lens = None
            ctx["lens"] = None
        ctx["unpad"] = unpad
        x32 = None
        if inputs_embeds is not None:
            x = inputs_embeds.to(device=dev, dtype=dt).reshape(T, d).contiguous()
        elif r32:  # the decoder's fp32 stream starts from the projector's fp32 output and the widened embedding rows
            x = None
            x32 = O.embed_splice_fwd_f32(input_ids.view(-1), src.view(-1) if src is not None else None,
                                         A.view("model.embed_tokens.weight", shape=(cfg.vocab_size, d)), feats)
        else:
            x = O.embed_splice_fwd(input_ids.view(-1), src.view(-1) if src is not None else None,
                                   A.view("model.embed_tokens.weight", shape=(cfg.vocab_size, d)), feats)
        ctx["ids"] = input_ids
        del feats
        # ---- decoder ----
        xs, saves = [], []
        fp8_train = fp8 == "train"
        if fp8 and (d % 128 or cfg.intermediate_size % 128):
            raise RuntimeError("the fp8 GEMM path needs hidden and intermediate sizes that are multiples of 128")
        if fp8 and not fp8_train:
            if want_grad:
                raise RuntimeError("model.fp8_forward is the inference form (forward only); set model.fp8_training = True for the fp8 training step")
            F8 = getattr(self, "_fp8_fwd", None) or self.quantize_forward_weights()
        if r32:
            if x32 is None:
                x32 = O.convert(x, torch.empty(x.shape, dtype=torch.float32, device=dev))
            for li, W in enumerate(self.llama):
                x16, sv = self._llama_layer_fwd_r32(W, x32, B, S, lens, keep=want_grad and self.save_activations,
                                                    kv_out=(kv_cache.k[li], kv_cache.v[li]) if kv_cache is not None else None, unpad=unpad,
                                                    need_x16=want_grad)
                if want_grad:
                    xs.append(x16)
                saves.append(sv)
        for li, W in enumerate(self.llama if not r32 else ()):
            if fp8_train:
                if kv_cache is not None:
                    raise RuntimeError("prefill runs the 16-bit or fp8-forward path")
                if want_grad:
                    xs.append(x)
                x, sv = self._llama_layer_fwd_fp8_train(W, li, x, B, S, lens, keep=want_grad and self.save_activations, unpad=unpad)
                saves.append(sv)
                continue
            if fp8:
                x = self._llama_layer_fwd_fp8(W, F8[li], x, B, S, lens,
                                              kv_out=(kv_cache.k[li], kv_cache.v[li]) if kv_cache is not None else None, unpad=unpad)
                saves.append(None)
                continue
            if want_grad:
                xs.append(x)
            x, sv = self._llama_layer_fwd(W, x, B, S, lens, keep=want_grad and self.save_activations,
                                          kv_out=(kv_cache.k[li], kv_cache.v[li]) if kv_cache is not None else None, unpad=unpad)
            saves.append(sv)
        if r32:
            hn, x = O.norm_fwd_f32in(x32, A.view("model.norm.weight"), cfg.rms_norm_eps, want_x16=want_grad)
            del x32
        else:
            hn = O.rmsnorm_fwd(x, A.view("model.norm.weight"), cfg.rms_norm_eps)
        ctx.update(xs=xs, saves=saves, x_last=x if want_grad else None)
        # ---- lm_head + shifted CE (llama_mmgpt.py:87-100) ----
        V = cfg.vocab_size
        Vpad = _ru(V, 64)
        wlm = A.view("lm_head.weight", numel=Vpad * d, shape=(Vpad, d))
        if last_only:  # prefill of generate(): only each sequence's last valid position feeds the sampler
            last = (lens.to(torch.int64) - 1) if lens is not None else torch.full((B,), S - 1, dtype=torch.int64, device=dev)
            rows = O.gather_rows(hn, torch.arange(B, device=dev) * S + last.clamp_min(0))
            return None, O.gemv(rows, wlm, out_f32=True, n=V), ctx
        # fp8 training step: the head's three GEMMs run on the scaled-fp8 MFMA too (BASELINE cfg 5: every Linear) when the geometry
        # allows it (contractions over d, V and T in whole 128-blocks); the logits stay fp32
        fp8_head = bool(fp8_train and self.fp8_head and not last_only and d % 128 == 0)
        ctx["fp8_head"] = fp8_head
        self.last_fp8 = dict(decoder=bool(fp8_train), tower=bool(ctx.get("fp8_tower")), head=fp8_head)  # which parts of this forward ran on fp8 (tests, bench)
        if fp8_head:
            hn8 = O.quant_fp8_rows(hn)
            logits = O.gemm_fp8(hn8, self.fp8_head_weights()["w"], out=torch.empty(T, Vpad, dtype=torch.float32, device=dev), dt16=dt)
            ctx["hn_scales"] = hn8[1] if want_grad else None
            del hn8
        else:
            logits = O.gemm_nt(hn, wlm, out_f32=True)  # [T, Vpad] fp32
        loss = None
        if labels is not None:
            row_loss, lse, out = O.ce_fwd(logits, labels, V)
            loss = out[2]
            ctx.update(labels=labels, ce_lse=lse, ce_out=out)
            if want_grad and self.sparse_head and not fp8_head and T % 64 == 0:
                # scored rows: position (b, s) iff s < S - 1 and labels[b, s + 1] != -100; row tables + count now, the count is read in backward
                sup = torch.zeros(B, S, dtype=torch.uint8, device=dev)
                sup[:, :-1] = labels[:, 1:] != -100
                rows_f, rows_i, cnt = O.mask_unpad_index(sup.view(1, T))
                # one pinned word PER forward context (torch's caching host allocator recycles it when the context dies): any number of
                # grad-enabled forwards may be outstanding before their backwards run (losses of many micro-batches summed, one .backward())
                slot = torch.empty(1, dtype=torch.int32, pin_memory=True)
                slot.copy_(cnt, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                ctx["scored_rows"] = (rows_f, rows_i, slot, ev)
        ctx.update(hn=hn if want_grad else None, logits=logits if want_grad else None)
        lg = logits.view(B, S, Vpad)[:, :, :V]
        return loss, lg, ctx

    def backward(self, ctx, gscale=1.0):
        """d(loss)/d(params) * gscale into the gradient arena (param.grad views)."""
        m = self.model
        cfg = m.config
        A = self.arena
        B, S = ctx["B"], ctx["S"]
        T, d, V = B * S, cfg.hidden_size, cfg.vocab_size
        Vpad = _ru(V, 64)
        Tpad = _ru(T, 64)
        dt = A.flat.dtype
        fresh = A.ensure_grads()
        if self.on_backward_begin is not None:
            self.on_backward_begin(fresh)
        acc = not fresh
        lens = ctx["lens"]
        # Every trainable bucket is reported through _ready() on EVERY backward, in one fixed order (head, decoder layers
        # L-1..0, embedding, projector, tower layers, tower embeddings), whatever this rank's batch contained: the
        # data-parallel all-reduce sequence is then identical on all ranks (merlin_amd/dp.py).
        # ---- head ----
        wlm = A.view("lm_head.weight", numel=Vpad * d, shape=(Vpad, d))
        sparse = None
        if ctx.get("scored_rows") is not None and not ctx.get("fp8_head"):
            rows_f, rows_i, host, ev = ctx["scored_rows"]
            ev.synchronize()  # (recorded in the forward: long done)
            n = int(host[0])
            npad = _ru(max(n, 1), 256)
            if 2 * npad <= T:
                sparse = (rows_f[:npad], rows_i)
        if sparse is not None:
            rows_f, rows_i = sparse
            npad = rows_f.numel()
            dl_c = O.ce_bwd_rows(ctx["logits"], ctx["labels"], ctx["ce_lse"], ctx["ce_out"], rows_f, V, Vpad, float(gscale), dt)  # [npad, Vpad], zero rows behind the count
            ctx["logits"] = None
            hn_c = O.gather_rows2d(ctx["hn"], rows_f, torch.empty(npad, d, dtype=dt, device=dl_c.device))
            dhn_c = O.gemm_nt(dl_c, wlm, b_t=True)  # [npad, d]
            if self._trainable("lm_head.weight"):
                off = A.offset["lm_head.weight"]
                O.gemm_nt(dl_c, hn_c, a_t=True, b_t=True, out=A.gflat[off: off + Vpad * d].view(Vpad, d), accum=acc)
            dhn = O.gather_rows2d(dhn_c, rows_i, torch.empty(T, d, dtype=dt, device=dl_c.device))  # un-scored rows: exact zeros
            del dl_c, hn_c, dhn_c
            dlogits = None
        else:
            dlogits = O.ce_bwd(ctx["logits"], ctx["labels"], ctx["ce_lse"], ctx["ce_out"], V, Vpad, float(gscale), dt)
            ctx["logits"] = None
        if sparse is not None:
            pass
        elif ctx.get("fp8_head"):
            train_head = self._trainable("lm_head.weight")
            V128 = _ru(Vpad, 128)  # the dgrad contracts over the (padded) vocabulary in whole 128-blocks: zero columns behind Vpad
            dl8, dlT8 = O.quant_fp8_both(dlogits, c_pad=V128) if train_head else (O.quant_fp8_rows(dlogits, k_pad=V128), None)
            dhn = O.gemm_fp8(dl8, self.fp8_head_weights()["wT"], out_dtype=dt)  # [T, d]
            if train_head:  # rows [V, Vpad) of the padded gradient block receive exact zeros (dlogits' pad columns are zero)
                off = A.offset["lm_head.weight"]
                self._wgrad_fp8(dlT8, ctx["hn"], ctx["hn_scales"], A.gflat[off: off + Vpad * d].view(Vpad, d), fresh)
            del dl8, dlT8
        else:
            dhn = O.gemm_nt(dlogits, wlm, b_t=True)  # [T, d]; B = W^T [d, Vpad]
        if sparse is None and self._trainable("lm_head.weight") and not ctx.get("fp8_head"):
            if T % 64 == 0:
                # rows [V, Vpad) of the padded gradient block receive exact zeros (dlogits' pad columns are zero)
                off = A.offset["lm_head.weight"]
                O.gemm_nt(dlogits, ctx["hn"], a_t=True, b_t=True, out=A.gflat[off: off + Vpad * d].view(Vpad, d), accum=acc)
            else:
                dlT = O.transpose16(dlogits, r_pad=Tpad)  # [Vpad, Tpad]
                hnT = O.transpose16(ctx["hn"], r_pad=Tpad)
                O.gemm_nt(dlT[:V], hnT, out=A.gview("lm_head.weight"), accum=acc)
                del dlT, hnT
        del dlogits
        dx = O.rmsnorm_bwd(ctx["x_last"], A.view("model.norm.weight"), dhn, cfg.rms_norm_eps,
                           dw_out=A.gview("model.norm.weight") if self._trainable("model.norm.weight") else None, dw_accumulate=acc)
        self._ready(["lm_head.weight", "model.norm.weight"])
        ctx["hn"] = ctx["x_last"] = None
        # ---- decoder ----
        for i in reversed(range(len(self.llama))):
            if ctx.get("fp8_train"):
                dx = self._llama_layer_bwd_fp8(self.llama[i], i, ctx["xs"][i], dx, B, S, lens, ctx["saves"][i], fresh, unpad=ctx.get("unpad"))
            else:
                dx = self._llama_layer_bwd(self.llama[i], ctx["xs"][i], dx, B, S, lens, ctx["saves"][i], fresh, unpad=ctx.get("unpad"),
                                           rows=sparse if (i == len(self.llama) - 1 and self.sparse_last_layer) else None)
            ctx["xs"][i] = None
            ctx["saves"][i] = None
        # ---- embedding + splice ----
        src = ctx.get("src")
        ids = ctx["ids"]
        inner = m.get_model()
        tower = getattr(inner, "vision_tower", None)
        proj = getattr(inner, "projector", None)
        wname, bname = "model.projector.projector.weight", "model.projector.projector.bias"
        emb_train = self._trainable("model.embed_tokens.weight")
        proj_train = proj is not None and any(p.requires_grad for p in proj.parameters())
        tower_train = tower is not None and not tower.freeze_vision_tower and any(p.requires_grad for p in tower.parameters())
        need_feats = src is not None and (ctx["train_tower"] or proj_train)
        dfeats = torch.zeros(ctx["n_feat_rows"], d, dtype=dt, device=dx.device) if need_feats else None
        dembed32 = torch.zeros(V, d, dtype=torch.float32, device=dx.device) if (emb_train and ids is not None) else None
        if ids is not None and (need_feats or emb_train):
            O.embed_splice_bwd(ids.view(-1), src.view(-1) if src is not None else None, dx, dfeats, dembed32)
        if emb_train:
            g = A.gview("model.embed_tokens.weight")
            if ids is None:  # inputs_embeds path: no token rows were looked up, the embedding gradient of this pass is zero
                if fresh:
                    g.zero_()
            elif acc:
                tmp = torch.empty_like(g)
                O.convert(dembed32, tmp)
                O.add(g, tmp, out=g)
            else:
                O.convert(dembed32, g)
            self._ready(["model.embed_tokens.weight"])
        del dx
        if need_feats:
            dxt = self.projector_bwd(ctx, dfeats, fresh)
            if ctx["train_tower"]:
                self.tower_bwd(ctx, dxt, fresh)
        elif src is None:
            # a batch without image tokens (text-only step, decode-shaped input): image-side gradients of this pass are zero.
            # The reference keeps those parameters in the graph with `0 * projector(dummy_feature)` (base_mmgpt.py:109-113);
            # here their buckets are zero-filled (first micro-step only) and still reported, in the usual order.
            if proj_train:
                self._untouched([wname, bname], fresh)
            if tower_train:
                for i in reversed(range(tower.layers_used)):
                    self._untouched(self.vit[i].names, fresh)
                self._untouched([n for n in A.names if n.startswith(VT + "embeddings.") or n.startswith(VT + "pre_layrnorm")], fresh)
        # trainable parameters no backward ever reaches (CLIP layers past select_layer, post_layernorm; a tower whose
        # freeze flag is set while its tensors still require grad): zero, never reported (identically on every rank)
        if fresh and tower is not None:
            L = tower.layers_used
            for n in A.names:
                if not n.startswith(VT) or not A.params[n].requires_grad:
                    continue
                dead = n.startswith(VT + "post_layernorm") or any(n.startswith(VT + f"encoder.layers.{i}.") for i in range(L, tower.config.num_hidden_layers))
                if dead or not tower_train:
                    A.gview(n).zero_()
        self._ready(None)  # end of backward

    # ------------------------------------------------------------------------------------------
    # KV-cache decode (llama_mmgpt.py:114-134, HF LlamaAttention with past_key_values)
    # ------------------------------------------------------------------------------------------
    class KVCache:
        """Per-layer rotated keys and values, [B, Smax, H*D] each, plus the valid length of every sequence."""

        def __init__(self, n_layers, B, Smax, d, dtype, device):
            self.k = [torch.zeros(B, Smax, d, dtype=dtype, device=device) for _ in range(n_layers)]
            self.v = [torch.zeros(B, Smax, d, dtype=dtype, device=device) for _ in range(n_layers)]
            self.lens = torch.zeros(B, dtype=torch.int32, device=device)
            # rotary position of the next token when it differs from its cache row (prompts with padding in front of / inside
            # them keep only their valid keys, positions stay absolute as HF numbers them); None: == lens
            self.rpos = None
            self.B, self.Smax = B, Smax
            self.k_alt = self.v_alt = None

    def prefill(self, input_ids, attention_mask, images, max_new_tokens, inputs_embeds=None):
        """Full forward over the prompt that also fills a KV cache; returns (logits fp32 [B, V] at each sequence's last
        valid position, cache).  Right-padded prompts (attention_mask) decode from their own length."""
        self.ensure_arena()
        cfg = self.model.config
        B, S = (input_ids.shape if input_ids is not None else inputs_embeds.shape[:2])
        A = self.arena
        cache = HipEngine.KVCache(len(self.llama), B, S + max_new_tokens, cfg.hidden_size, A.flat.dtype, A.flat.device)
        self._rope_table(S + max_new_tokens, A.flat.device)
        _, logits, ctx = self.forward(input_ids, attention_mask, None, images, inputs_embeds=inputs_embeds, kv_cache=cache,
                                      last_only=True)
        lens = ctx["lens"]
        if ctx.get("unpad") is not None:
            # general mask (left padding / holes): the cache holds the valid keys only; the next token's rotary position is the
            # padded prompt length, as LlamaModel numbers it when generate() passes no position_ids (llama_mmgpt.py:114-134)
            cache.lens.copy_(ctx["unpad"][2])
            cache.rpos = torch.full((B,), S, dtype=torch.int32, device=A.flat.device)
        else:
            cache.lens.copy_(lens if lens is not None else torch.full((B,), S, dtype=torch.int32, device=A.flat.device))
        return logits, cache

    def expand_cache(self, cache, rows):
        """New cache whose row i is a copy of `cache` row rows[i] (int64 on the device): a prefilled batch expanded to
        num_beams rows per prompt (the reference: inputs_embeds.repeat_interleave(5), base_mmgpt.py:162-163)."""
        A = self.arena
        n = int(rows.numel())
        Smax, d = cache.Smax, cache.k[0].shape[-1]
        out = HipEngine.KVCache(0, n, Smax, d, A.flat.dtype, A.flat.device)
        for li in range(len(cache.k)):
            for src_l, dst_l in ((cache.k, out.k), (cache.v, out.v)):
                dst = torch.empty(n, Smax, d, dtype=A.flat.dtype, device=A.flat.device)
                O.gather_rows2d(src_l[li].view(cache.B, Smax * d), rows, dst.view(n, Smax * d))
                dst_l.append(dst)
        out.lens = cache.lens.index_select(0, rows).contiguous()  # B int32 values, once per generate() call
        out.rpos = cache.rpos.index_select(0, rows).contiguous() if cache.rpos is not None else None
        return out

    def reorder_cache(self, cache, beam_idx, n_valid):
        """cache row i <- cache row beam_idx[i] for the first n_valid positions (HF `_reorder_cache` under beam search):
        one gather kernel per layer and tensor into a second buffer set, then the sets swap."""
        d = cache.k[0].shape[-1]
        if getattr(cache, "k_alt", None) is None:
            cache.k_alt = [torch.empty_like(t) for t in cache.k]
            cache.v_alt = [torch.empty_like(t) for t in cache.v]
        n = cache.B
        for li in range(len(cache.k)):
            O.gather_rows2d(cache.k[li].view(n, -1), beam_idx, cache.k_alt[li].view(n, -1), cols=n_valid * d)
            O.gather_rows2d(cache.v[li].view(n, -1), beam_idx, cache.v_alt[li].view(n, -1), cols=n_valid * d)
        cache.k, cache.k_alt = cache.k_alt, cache.k
        cache.v, cache.v_alt = cache.v_alt, cache.v

    def quantize_decode_weights(self):
        """fp8 (OCP e4m3, one scale per 128 k) copies of the decoder's Linear weights for the decode step (BASELINE cfg 5's
        weight format; activations stay 16-bit).  ~half the bytes per token.  Re-run after the weights change."""
        self.ensure_arena()
        A = self.arena
        cfg = self.model.config
        V, d = cfg.vocab_size, cfg.hidden_size
        q = []
        for W in self.llama:
            q.append(dict(wqkv=O.quant_fp8_b128(W.wqkv), wo=O.quant_fp8_b128(W.wo), wgu=O.quant_fp8_b128(W.wgu), wd=O.quant_fp8_b128(W.wd)))
        Vpad = _ru(V, 64)
        wlm = A.view("lm_head.weight", numel=Vpad * d, shape=(Vpad, d))
        self._fp8 = dict(layers=q, lm_head=O.quant_fp8_b128(wlm[:V]))
        return self._fp8

    def decode_step(self, tokens, cache, fp8=False):
        """One new token per sequence (tokens int64 [B]) at position cache.lens[b]; returns logits fp32 [B, V] and
        advances the cache.  Every op is an HBM-bound kernel: weights and cache are streamed exactly once.
        fp8=True uses the fp8 weight copies of quantize_decode_weights()."""
        if fp8:
            return self._decode_step_fp8(tokens, cache)
        cfg = self.model.config
        A = self.arena
        d, H, D, V = cfg.hidden_size, cfg.num_attention_heads, head_dim_of(cfg), cfg.vocab_size
        eps = cfg.rms_norm_eps
        emb = A.view("model.embed_tokens.weight", shape=(V, d))
        x = O.gather_rows(emb, tokens.to(A.flat.device).view(-1))
        pos = cache.lens
        lens1 = pos + 1
        for li, W in enumerate(self.llama):
            # input_layernorm + q|k|v projection + RoPE + K/V append: one launch
            qkv = O.gemv_qkv_rope(x, W.ln1, eps, W.wqkv, self.rope, pos, cache.k[li], cache.v[li], H, D, rope_pos=cache.rpos)
            o = O.attn_decode(qkv[:, :d], cache.k[li], cache.v[li], lens1, H, D)
            x2 = O.gemv(o, W.wo, resid=x)
            act = O.gemv_norm(x2, W.ln2, eps, W.wgu, swiglu=True)  # post_attention_layernorm + gate|up + SwiGLU: one launch
            x = O.gemv(act, W.wd, resid=x2)
        hn = O.rmsnorm_fwd(x, A.view("model.norm.weight"), eps)
        Vpad = _ru(V, 64)
        wlm = A.view("lm_head.weight", numel=Vpad * d, shape=(Vpad, d))
        logits = O.gemv(hn, wlm, out_f32=True, n=V)
        cache.lens.add_(1)  # in place (after every kernel that read it as `pos`): the captured graph sees the same buffer
        if cache.rpos is not None:
            cache.rpos.add_(1)
        return logits

    def _decode_step_fp8(self, tokens, cache):
        cfg = self.model.config
        A = self.arena
        F8 = getattr(self, "_fp8", None) or self.quantize_decode_weights()
        d, H, D, V = cfg.hidden_size, cfg.num_attention_heads, head_dim_of(cfg), cfg.vocab_size
        eps = cfg.rms_norm_eps
        emb = A.view("model.embed_tokens.weight", shape=(V, d))
        x = O.gather_rows(emb, tokens.to(A.flat.device).view(-1))
        pos = cache.lens
        lens1 = pos + 1
        for li, W in enumerate(self.llama):
            Q = F8["layers"][li]
            qkv = O.gemv_qkv_rope(x, W.ln1, eps, Q["wqkv"], self.rope, pos, cache.k[li], cache.v[li], H, D, rope_pos=cache.rpos)
            o = O.attn_decode(qkv[:, :d], cache.k[li], cache.v[li], lens1, H, D)
            x2 = O.gemv_fp8w(o, Q["wo"], resid=x)
            act = O.gemv_fp8w_norm(x2, W.ln2, eps, Q["wgu"], swiglu=True)
            x = O.gemv_fp8w(act, Q["wd"], resid=x2)
        hn = O.rmsnorm_fwd(x, A.view("model.norm.weight"), eps)
        logits = O.gemv_fp8w(hn, F8["lm_head"], out_f32=True)
        cache.lens.add_(1)
        if cache.rpos is not None:
            cache.rpos.add_(1)
        return logits

    def capture_decode_graph(self, cache, fp8=False):
        """Capture one decode step (≈300 launches) into a HIP graph bound to `cache`: returns (graph, token buffer int64 [B],
        logits buffer fp32 [B, V]).  Positions live in cache.lens on the device and advance inside the graph, so every
        replay is the next token.  One eager warm-up step runs first (function attributes, symbol look-ups and allocator
        warm-up must not happen during capture); it writes only the cache rows the first real step rewrites."""
        dev = self.arena.flat.device
        tok = torch.zeros(cache.B, dtype=torch.int64, device=dev)
        keep = cache.lens.clone()
        keep_r = cache.rpos.clone() if cache.rpos is not None else None
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self.decode_step(tok, cache, fp8=fp8)
        torch.cuda.current_stream(dev).wait_stream(side)
        cache.lens.copy_(keep)
        if keep_r is not None:
            cache.rpos.copy_(keep_r)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            logits = self.decode_step(tok, cache, fp8=fp8)
        cache.lens.copy_(keep)  # capture does not execute, but keep the invariant explicit
        if keep_r is not None:
            cache.rpos.copy_(keep_r)
        return g, tok, logits

    # standalone sub-module calls (reference module surface; not used by the fused forward)
    def tower_forward_public(self, images):
        self.ensure_arena()
        tower = self.model.get_model().vision_tower
        x, N, S = self.tower(images, None)
        vd = x.shape[1]
        x3 = x.view(N, S, vd)
        feats = x3 if tower.select_feature == "cls_patch" else x3[:, 1:]
        feats = feats.to(images[0].dtype)
        return torch.split(feats, [im.shape[0] for im in images], dim=0)

    def projector_forward_public(self, features):
        """mlp_projector.py:19-23 / conv_projector.py:23-39 as a standalone call: list of [n_i, P, C] -> list of [n_i, P', d]."""
        self.ensure_arena()
        A = self.arena
        proj = self.model.get_model().projector
        wname, bname = "model.projector.projector.weight", "model.projector.projector.bias"
        out = []
        for f in features:
            if hasattr(proj, "conv_stride"):
                if f.dim() == 1:  # conv_projector.py:27-28: the 1-D dummy feature is tiled to 256 tokens
                    f = f.view(1, 1, -1).repeat(1, 256, 1)
                n, P, C = f.shape
                G = int(math.sqrt(P))
                x = f.to(A.flat.dtype).reshape(n * P, C).contiguous()
                cols = O.conv3x3_cols(x, n, G, C, proj.conv_stride, P, 0)
                w = A.view(wname, shape=(A.params[wname].shape[0], C * 9))
                y = O.gemm_nt(cols, w, bias=A.view(bname))
                out.append(y.view(n, -1, y.shape[-1]))
                continue
            f2 = f.to(A.flat.dtype).reshape(-1, f.shape[-1]).contiguous()
            y = O.gemm_nt(f2, A.view(wname), bias=A.view(bname))
            out.append(y.view(*f.shape[:-1], -1))
        return out
