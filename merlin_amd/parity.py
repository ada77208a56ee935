"""fp32-store parity forward (SURVEY §8d cfg 2).  `model.engine.parity_fp32 = True` routes
MMGPTLlamaForCausalLM.forward (no gradients) through this module: the same computation as merlin_amd/model/engine.py, but every
activation lives in HBM as fp32, so the comparison with the reference's fp32 CPU path (BASELINE: "logits within 1e-3 rel")
measures the kernels and not 16-bit activation storage.  Linear layers run on the production bf16 MFMA GEMM through an exact
3-term bf16 split of the fp32 activations (include/merlin_hip.h: mh_p32_*); weights are the model's own (exactly representable
in bf16).  Forward only, not a performance path, never taken unless asked for."""
from __future__ import annotations

import math

import torch

from . import _lib as L
from . import ops as O
from ._lib import f32, i32, i64, p
from .model.config import head_dim_of
from .ops import _stream

VT = "model.vision_tower.vision_tower.vision_model."


def _ru(x, m):
    return (x + m - 1) // m * m


def _split3(x32):
    n = x32.numel()
    hml = torch.empty(3, *x32.shape, dtype=torch.bfloat16, device=x32.device)
    L.check(L.lib().mh_p32_split3(p(x32), p(hml[0]), p(hml[1]), p(hml[2]), i64(n), _stream()), "mh_p32_split3")
    return hml[0], hml[1], hml[2]


def linear(x32, w, bias=None):
    """fp32 [M, K] x bf16 [N, K]^T (+ bf16 bias) -> fp32 [M, N]: smallest term first, fp32 accumulate in the GEMM epilogue."""
    assert x32.dtype == torch.float32 and w.dtype == torch.bfloat16 and x32.is_contiguous()
    hi, mid, lo = _split3(x32)
    out = O.gemm_nt(lo, w, out_f32=True)
    O.gemm_nt(mid, w, out=out, accum=True)
    O.gemm_nt(hi, w, out=out, accum=True, bias=bias)
    return out


def rmsnorm(x, w, eps):
    y = torch.empty_like(x)
    L.check(L.lib().mh_p32_rmsnorm(p(x), p(w), p(y), i32(x.shape[0]), i32(x.shape[1]), f32(eps), _stream()), "mh_p32_rmsnorm")
    return y


def layernorm(x, w, b, eps):
    y = torch.empty_like(x)
    L.check(L.lib().mh_p32_layernorm(p(x), p(w), p(b), p(y), i32(x.shape[0]), i32(x.shape[1]), f32(eps), _stream()), "mh_p32_layernorm")
    return y


def _ew(a, b, op, out_shape=None, ff=0):
    y = torch.empty(out_shape if out_shape is not None else a.shape, dtype=torch.float32, device=a.device)
    L.check(L.lib().mh_p32_elementwise(p(a), p(b), p(y), i64(y.numel()), i32(op), i32(ff), _stream()), "mh_p32_elementwise")
    return y


def attention(qkv, B, S, H, D, causal, lens):
    d = H * D
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    o = torch.empty(B * S, d, dtype=torch.float32, device=qkv.device)
    L.check(L.lib().mh_p32_attention(p(q), i64(q.stride(0)), p(k), i64(k.stride(0)), p(v), i64(v.stride(0)), p(o), i64(d), p(lens), i32(B), i32(S),
                                     i32(H), i32(D), i32(int(causal)), _stream()), "mh_p32_attention")
    return o


class _W:
    """bf16 views of the parameter arena (the arena itself for bf16 models, an exact converted copy for fp16 models)."""

    def __init__(self, engine):
        A = engine.arena
        self.A = A
        if A.flat.dtype == torch.bfloat16:
            self.flat = A.flat
        else:
            self.flat = engine._derive("parity_bf16_arena", lambda: O.convert(A.flat, torch.empty_like(A.flat, dtype=torch.bfloat16)))

    def view(self, name, shape=None, numel=None):
        A = self.A
        n = A.params[name].numel() if numel is None else numel
        v = self.flat[A.offset[name]: A.offset[name] + n]
        return v.view(shape if shape is not None else (A.params[name].shape if numel is None else (n,)))

    def span(self, first, last, shape):
        A = self.A
        i0, i1 = A.names.index(first), A.names.index(last)
        n = sum(A.params[x].numel() for x in A.names[i0: i1 + 1])
        return self.flat[A.offset[first]: A.offset[first] + n].view(shape)


@torch.no_grad()
def forward(engine, input_ids, attention_mask, labels, images):
    """-> (loss fp32 scalar | None, logits fp32 [B, S, V])."""
    m = engine.model
    cfg = m.config
    A = engine.ensure_arena()
    W = _W(engine)
    dev = A.flat.device
    inner = m.get_model()
    tower = getattr(inner, "vision_tower", None)
    B, S = input_ids.shape
    T, d, ff, H, D = B * S, cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, head_dim_of(cfg)
    rope = engine._rope_table(S, dev)
    input_ids = input_ids.to(dev).contiguous()
    lens = am = None
    if attention_mask is not None:
        am = attention_mask.to(dev)
        am = (am if am.dtype == torch.bool else am != 0).contiguous()
        lens = O.mask_lens(am)
    if labels is not None:
        labels = labels.to(dev).contiguous()
    use_images = tower is not None and images is not None and S != 1
    src = engine.validate_and_index(input_ids, labels, am, lens, images, use_images)
    feats = None
    if use_images:
        vc = tower.config
        G = vc.image_size // vc.patch_size
        G2, Sv, vd = G * G, G * G + 1, vc.hidden_size
        K = 3 * vc.patch_size * vc.patch_size
        Kpad = _ru(K, 64)
        pix = torch.cat([im.to(dev).float() for im in images], dim=0).contiguous()
        N = pix.shape[0]
        cols = torch.empty(N * Sv, Kpad, dtype=torch.float32, device=dev)
        L.check(L.lib().mh_p32_im2col(p(pix), p(cols), i32(N), i32(vc.image_size), i32(vc.patch_size), i32(Kpad), i32(Sv), i32(1), _stream()), "mh_p32_im2col")
        wpad = torch.zeros(vd, Kpad, dtype=torch.bfloat16, device=dev)
        O.copy2d(W.view(VT + "embeddings.patch_embedding.weight", shape=(vd, K)), wpad)
        patch = linear(cols, wpad)
        x = torch.empty(N * Sv, vd, dtype=torch.float32, device=dev)
        L.check(L.lib().mh_p32_vit_assemble(p(patch), p(W.view(VT + "embeddings.class_embedding")), p(W.view(VT + "embeddings.position_embedding.weight")),
                                            p(x), i32(N), i32(G2), i32(vd), _stream()), "mh_p32_vit_assemble")
        eps = vc.layer_norm_eps
        x = layernorm(x, W.view(VT + "pre_layrnorm.weight"), W.view(VT + "pre_layrnorm.bias"), eps)
        Hv = vc.num_attention_heads
        for i in range(tower.layers_used):
            q = VT + f"encoder.layers.{i}."
            h1 = layernorm(x, W.view(q + "layer_norm1.weight"), W.view(q + "layer_norm1.bias"), eps)
            qkv = linear(h1, W.span(q + "self_attn.q_proj.weight", q + "self_attn.v_proj.weight", (3 * vd, vd)),
                         W.span(q + "self_attn.q_proj.bias", q + "self_attn.v_proj.bias", (3 * vd,)))
            o = attention(qkv, N, Sv, Hv, vd // Hv, False, None)
            x2 = _ew(linear(o, W.view(q + "self_attn.out_proj.weight"), W.view(q + "self_attn.out_proj.bias")), x, 0)
            h2 = layernorm(x2, W.view(q + "layer_norm2.weight"), W.view(q + "layer_norm2.bias"), eps)
            a = _ew(linear(h2, W.view(q + "mlp.fc1.weight"), W.view(q + "mlp.fc1.bias")), None, 2)
            x = _ew(linear(a, W.view(q + "mlp.fc2.weight"), W.view(q + "mlp.fc2.bias")), x2, 0)
        proj = inner.projector
        wn, bn = "model.projector.projector.weight", "model.projector.projector.bias"
        if hasattr(proj, "conv_stride"):
            Go = (G + 2 - 3) // proj.conv_stride + 1
            c32 = torch.empty(N * Go * Go, vd * 9, dtype=torch.float32, device=dev)
            L.check(L.lib().mh_p32_conv3x3_cols(p(x), p(c32), i32(N), i32(G), i32(vd), i32(proj.conv_stride), i32(Sv), i32(1), _stream()), "mh_p32_conv3x3_cols")
            feats = linear(c32, W.view(wn, shape=(A.params[wn].shape[0], vd * 9)), W.view(bn))
        else:
            feats = linear(x, W.view(wn), W.view(bn))
        assert engine._splice_geometry()[0] * N == feats.shape[0]
    engine._check_errors()
    x = torch.empty(T, d, dtype=torch.float32, device=dev)
    L.check(L.lib().mh_p32_embed_splice(p(input_ids), p(src), p(W.view("model.embed_tokens.weight")), p(feats), p(x), i64(T), i32(d), _stream()),
            "mh_p32_embed_splice")
    eps = cfg.rms_norm_eps
    for i in range(cfg.num_hidden_layers):
        q = f"model.layers.{i}."
        h1 = rmsnorm(x, W.view(q + "input_layernorm.weight"), eps)
        qkv = linear(h1, W.span(q + "self_attn.q_proj.weight", q + "self_attn.v_proj.weight", (3 * d, d)))
        L.check(L.lib().mh_p32_rope(p(qkv), p(rope), i64(T), i32(S), i32(H), i32(D), _stream()), "mh_p32_rope")
        o = attention(qkv, B, S, H, D, True, lens)
        x2 = _ew(linear(o, W.view(q + "self_attn.o_proj.weight")), x, 0)
        h2 = rmsnorm(x2, W.view(q + "post_attention_layernorm.weight"), eps)
        gu = linear(h2, W.span(q + "mlp.gate_proj.weight", q + "mlp.up_proj.weight", (2 * ff, d)))
        act = _ew(gu, None, 1, out_shape=(T, ff), ff=ff)
        del gu
        x = _ew(linear(act, W.view(q + "mlp.down_proj.weight")), x2, 0)
    hn = rmsnorm(x, W.view("model.norm.weight"), eps)
    V = cfg.vocab_size
    Vpad = _ru(V, 64)
    logits = linear(hn, W.view("lm_head.weight", numel=Vpad * d, shape=(Vpad, d)))
    loss = None
    if labels is not None:
        loss = O.ce_fwd(logits, labels, V)[2][2]
    return loss, logits.view(B, S, Vpad)[:, :, :V]
