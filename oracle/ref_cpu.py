"""CPU oracle: a plain-torch fp32 restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under `merlin_amd/` may import this module; only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use it, and only as
the checker / the timed CPU baseline, never as a fallback for the HIP path.

Pinning: the reference (Ahnsun/merlin @ 2024_08_07) ships no tests or golden vectors for
this path, so this restatement is pinned against the reference itself:
`oracle/make_golden.py` imports the real `MMGPTLlamaForCausalLM` (CPU, fp32) in the build
container and stores its inputs/outputs under `tests/golden/`; `tests/test_oracle_golden.py`
asserts this file reproduces those outputs.

What is restated (reference file:line -> function here):
  mmgpt/model/mmgpt/llama_mmgpt.py:53-112        -> forward()            (splice -> LlamaModel -> lm_head -> shifted CE)
  mmgpt/model/mmgpt/base_mmgpt.py:18-21          -> encode_images()
  mmgpt/model/mmgpt/base_mmgpt.py:82-165         -> splice_image_features()
  mmgpt/model/vision_encoder/clip_encoder.py:64-82 -> clip_tower_forward()  (hidden_states[select_layer][:, 1:])
  mmgpt/model/projector/mlp_projector.py:19-23   -> projector_forward('mlp')
  mmgpt/model/projector/conv_projector.py:23-39  -> projector_forward('conv')
  mmgpt/utils/llama_flash_attn_monkey_patch.py:20-103 -> llama_attention() (causal, key-padding)
Third-party arithmetic the reference delegates to (not vendored in /root/reference; pinned
transformers==4.31.0 in pyproject.toml:22, 5.15.0 installed here, formulas identical):
  transformers CLIPVisionModel (embeddings, pre_layrnorm, encoder layers, quick_gelu)
  transformers LlamaModel (RMSNorm, rotate-half RoPE theta=1e4, SwiGLU MLP)

Parameters are a flat dict keyed by the reference's state-dict names.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F

IGNORE_INDEX = -100  # mmgpt/utils/constants.py:7

# ---- storage-rounding model (tests/test_parity_floor_gpu.py, tools/measure_parity.py) -------------------------------------------------
# None (default): the plain fp32 restatement, bit-identical to the reference goldens.  `rounding(dtype, stream=...)` makes this SAME
# fp32 forward round, to `dtype`, every tensor that a 16-bit-operand matrix unit has to take as an input - the activation operand of
# every Linear (weights from merlin_amd/weights.py are exactly representable), the rotated q / k, v and the softmax probabilities of
# both attentions - and, with stream=True, the residual streams after every add as well.  Accumulation, norms, softmax, RoPE, SwiGLU
# stay fp32.  The logits error of that run against the reference golden is the FLOOR of any single-pass 16-bit-operand implementation:
# what the HIP path's own error is held against.  outputs=True additionally rounds the GEMM outputs the HIP path STORES in 16 bits
# before an element-wise op consumes them (q | k before the rotation, gate | up before SwiGLU: the 8-wave kernel's staged epilogues work on
# the rounded values so that fused and unfused kernels agree bit for bit; the 4-wave kernel's forms work on the fp32 accumulators) and, with
# 16-bit streams, the 16-bit tensors at the start of the streams - an upper model of the HIP path's own storage format.
_ROUND = None


class rounding:
    def __init__(self, dtype, stream=False, outputs=False):
        self.cfg = (dtype, bool(stream), bool(outputs))

    def __enter__(self):
        global _ROUND
        self.old, _ROUND = _ROUND, self.cfg

    def __exit__(self, *a):
        global _ROUND
        _ROUND = self.old


def _q(x):
    """a matrix-unit operand"""
    return x if _ROUND is None else x.to(_ROUND[0]).to(x.dtype)


def _qs(x):
    """a residual stream"""
    return x if (_ROUND is None or not _ROUND[1]) else x.to(_ROUND[0]).to(x.dtype)


def _qo(x):
    """a GEMM output stored in 16 bits ahead of a fused element-wise op"""
    return x if (_ROUND is None or not _ROUND[2]) else x.to(_ROUND[0]).to(x.dtype)


def _q0(x):
    """a 16-bit tensor at the START of a residual stream (patch projection, assembled embeddings, pre-LN output, projector output): the HIP
    path keeps these in 16 bits only when the stream itself is 16-bit (with fp32 streams they are fp32 since round 4)"""
    return x if (_ROUND is None or not (_ROUND[1] and _ROUND[2])) else x.to(_ROUND[0]).to(x.dtype)


def _linear(x, w, b=None):
    return F.linear(_q(x), w, b)

VT = "model.vision_tower.vision_tower.vision_model."
PJ = "model.projector.projector."


@dataclass
class OracleConfig:
    # Llama (transformers LlamaConfig field names)
    vocab_size: int = 32003
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    rms_norm_eps: float = 1e-6
    rope_theta: float = 10000.0
    # CLIP vision tower (transformers CLIPVisionConfig field names, v_ prefix)
    v_hidden_size: int = 1024
    v_intermediate_size: int = 4096
    v_num_hidden_layers: int = 24
    v_num_attention_heads: int = 16
    v_image_size: int = 336
    v_patch_size: int = 14
    v_layer_norm_eps: float = 1e-5
    vision_select_layer: int = -2  # mmgpt/utils/arguments.py:14
    vision_select_feature: str = "patch"
    # projector (mmgpt/utils/arguments.py:10,17)
    projector: str = "mlp"
    conv_stride: int = 1
    # special tokens (base_mmgpt.py:55-63): <im_patch>=V, <im_start>=V+1, <im_end>=V+2
    im_patch_token: int = 32000
    im_start_token: int = 32001
    im_end_token: int = 32002

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def v_grid(self) -> int:
        return self.v_image_size // self.v_patch_size

    @property
    def num_patches(self) -> int:  # clip_encoder.py:105-107
        return (self.v_grid // self.conv_stride) ** 2

    @property
    def v_layers_used(self) -> int:
        n = self.v_num_hidden_layers + 1  # len(hidden_states)
        idx = self.vision_select_layer if self.vision_select_layer >= 0 else n + self.vision_select_layer
        return idx  # hidden_states[idx] = output after `idx` encoder layers


def param_shapes(cfg: OracleConfig) -> dict:
    """Reference state-dict key -> shape (SURVEY.md §8b 'State-dict key names')."""
    d, ff, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    vd, vff = cfg.v_hidden_size, cfg.v_intermediate_size
    s = {"model.embed_tokens.weight": (V, d), "model.norm.weight": (d,), "lm_head.weight": (V, d)}
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            s[p + f"self_attn.{n}.weight"] = (d, d)
        s[p + "mlp.gate_proj.weight"] = (ff, d)
        s[p + "mlp.up_proj.weight"] = (ff, d)
        s[p + "mlp.down_proj.weight"] = (d, ff)
        s[p + "input_layernorm.weight"] = (d,)
        s[p + "post_attention_layernorm.weight"] = (d,)
    npos = cfg.v_grid ** 2 + 1
    s[VT + "embeddings.class_embedding"] = (vd,)
    s[VT + "embeddings.patch_embedding.weight"] = (vd, 3, cfg.v_patch_size, cfg.v_patch_size)
    s[VT + "embeddings.position_embedding.weight"] = (npos, vd)
    for n in ("pre_layrnorm", "post_layernorm"):
        s[VT + n + ".weight"] = (vd,)
        s[VT + n + ".bias"] = (vd,)
    for i in range(cfg.v_num_hidden_layers):
        p = VT + f"encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"] = (vd, vd)
            s[p + f"self_attn.{n}.bias"] = (vd,)
        for n in ("layer_norm1", "layer_norm2"):
            s[p + n + ".weight"] = (vd,)
            s[p + n + ".bias"] = (vd,)
        s[p + "mlp.fc1.weight"] = (vff, vd)
        s[p + "mlp.fc1.bias"] = (vff,)
        s[p + "mlp.fc2.weight"] = (vd, vff)
        s[p + "mlp.fc2.bias"] = (vd,)
    if cfg.projector == "mlp":
        s[PJ + "weight"] = (d, vd)
    else:
        s[PJ + "weight"] = (d, vd, 3, 3)
    s[PJ + "bias"] = (d,)
    return s


# --------------------------------------------------------------------------------------
# CLIP vision tower (HF CLIPVisionModel arithmetic; called from clip_encoder.py:79)
# --------------------------------------------------------------------------------------
def clip_tower_forward(P: dict, cfg: OracleConfig, pixels: torch.Tensor) -> torch.Tensor:
    """pixels [N,3,H,W] -> hidden_states[select_layer][:, 1:]  ([N, grid^2, vd])."""
    N = pixels.shape[0]
    vd, nh = cfg.v_hidden_size, cfg.v_num_attention_heads
    hd = vd // nh
    x = _q0(F.conv2d(_q(pixels), P[VT + "embeddings.patch_embedding.weight"], stride=cfg.v_patch_size))
    x = x.flatten(2).transpose(1, 2)  # [N, grid^2, vd]
    cls = P[VT + "embeddings.class_embedding"].expand(N, 1, -1)
    x = _q0(torch.cat([cls, x], dim=1) + P[VT + "embeddings.position_embedding.weight"][None])
    x = _qs(F.layer_norm(x, (vd,), P[VT + "pre_layrnorm.weight"], P[VT + "pre_layrnorm.bias"], cfg.v_layer_norm_eps))
    for i in range(cfg.v_layers_used):
        p = VT + f"encoder.layers.{i}."
        r = x
        h = F.layer_norm(x, (vd,), P[p + "layer_norm1.weight"], P[p + "layer_norm1.bias"], cfg.v_layer_norm_eps)
        q = _linear(h, P[p + "self_attn.q_proj.weight"], P[p + "self_attn.q_proj.bias"])
        k = _linear(h, P[p + "self_attn.k_proj.weight"], P[p + "self_attn.k_proj.bias"])
        v = _linear(h, P[p + "self_attn.v_proj.weight"], P[p + "self_attn.v_proj.bias"])
        S = x.shape[1]
        q = _q(q).view(N, S, nh, hd).transpose(1, 2)
        k = _q(k).view(N, S, nh, hd).transpose(1, 2)
        v = _q(v).view(N, S, nh, hd).transpose(1, 2)
        att = torch.softmax((q @ k.transpose(-1, -2)) * (hd ** -0.5), dim=-1)
        o = (_q(att) @ v).transpose(1, 2).reshape(N, S, vd)
        x = _qs(r + _linear(o, P[p + "self_attn.out_proj.weight"], P[p + "self_attn.out_proj.bias"]))
        r = x
        h = F.layer_norm(x, (vd,), P[p + "layer_norm2.weight"], P[p + "layer_norm2.bias"], cfg.v_layer_norm_eps)
        h = _linear(h, P[p + "mlp.fc1.weight"], P[p + "mlp.fc1.bias"])
        h = h * torch.sigmoid(1.702 * h)  # quick_gelu
        x = _qs(r + _linear(h, P[p + "mlp.fc2.weight"], P[p + "mlp.fc2.bias"]))
    if cfg.vision_select_feature == "patch":  # clip_encoder.py:66-67
        x = x[:, 1:]
    return x


def projector_forward(P: dict, cfg: OracleConfig, feats: torch.Tensor) -> torch.Tensor:
    """[n, grid^2, vd] -> [n, P, d].  mlp_projector.py:19-23 / conv_projector.py:23-39."""
    if cfg.projector == "mlp":
        return _linear(feats, P[PJ + "weight"], P[PJ + "bias"])
    B, Pn, C = feats.shape
    HW = int(math.sqrt(Pn))
    f = feats.permute(0, 2, 1).reshape(B, C, HW, HW)
    y = F.conv2d(_q(f), P[PJ + "weight"], P[PJ + "bias"], stride=cfg.conv_stride, padding=1)
    return y.reshape(B, y.shape[1], -1).permute(0, 2, 1)


def encode_images(P: dict, cfg: OracleConfig, images: list) -> list:
    """base_mmgpt.py:18-21 + clip_encoder.py:74-82: cat -> one tower call -> split -> projector."""
    sizes = [im.shape[0] for im in images]
    feats = clip_tower_forward(P, cfg, torch.cat(list(images), dim=0).float())
    return [projector_forward(P, cfg, f) for f in torch.split(feats, sizes, dim=0)]


def splice_image_features(P: dict, cfg: OracleConfig, input_ids: torch.Tensor, image_features: list) -> torch.Tensor:
    """base_mmgpt.py:99-160: embed lookup, then rows p+1..p+P after each <im_start> at p are
    replaced by the sample's image features, in order (extra images silently ignored by zip)."""
    embeds = F.embedding(input_ids, P["model.embed_tokens.weight"])
    out = []
    for ids, emb, feats in zip(input_ids, embeds, image_features):
        if (ids == cfg.im_patch_token).sum() == 0:
            out.append(emb)  # text-only sample: reference adds 0 * projector(dummy)
            continue
        if (ids == cfg.im_start_token).sum() != (ids == cfg.im_end_token).sum():
            raise ValueError("The number of image start tokens and image end tokens should be the same.")
        starts = torch.where(ids == cfg.im_start_token)[0]
        for pos, f in zip(starts, feats):
            pos = int(pos)
            n = f.shape[0]
            if ids[pos + n + 1] != cfg.im_end_token:
                raise ValueError("The image end token should follow the image start token.")
            emb = torch.cat((emb[: pos + 1], f, emb[pos + n + 1:]), dim=0)
        out.append(emb)
    return torch.stack(out, dim=0)


# --------------------------------------------------------------------------------------
# Llama decoder (HF LlamaModel arithmetic)
# --------------------------------------------------------------------------------------
def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    v = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x.float() * torch.rsqrt(v + eps)).to(x.dtype)


def rope_tables(S: int, hd: int, theta: float):
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    fr = torch.outer(torch.arange(S, dtype=torch.float32), inv)
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def llama_attention(P, cfg, prefix, h, cos, sin, add_mask):
    B, S, d = h.shape
    nh, hd = cfg.num_attention_heads, cfg.head_dim
    q = _qo(_linear(h, P[prefix + "q_proj.weight"])).view(B, S, nh, hd).transpose(1, 2)
    k = _qo(_linear(h, P[prefix + "k_proj.weight"])).view(B, S, nh, hd).transpose(1, 2)
    v = _q(_linear(h, P[prefix + "v_proj.weight"])).view(B, S, nh, hd).transpose(1, 2)
    q = _q(q * cos + rotate_half(q) * sin)
    k = _q(k * cos + rotate_half(k) * sin)
    att = (q @ k.transpose(-1, -2)) / math.sqrt(hd) + add_mask
    att = torch.softmax(att, dim=-1, dtype=torch.float32)
    o = (_q(att) @ v).transpose(1, 2).reshape(B, S, d)
    return _linear(o, P[prefix + "o_proj.weight"])


def llama_forward(P: dict, cfg: OracleConfig, x: torch.Tensor, attention_mask=None) -> torch.Tensor:
    B, S, _ = x.shape
    cos, sin = rope_tables(S, cfg.head_dim, cfg.rope_theta)
    neg = torch.finfo(torch.float32).min
    causal = torch.full((S, S), neg).triu(1)
    add_mask = causal[None, None].expand(B, 1, S, S)
    if attention_mask is not None:
        pad = (~attention_mask.bool())[:, None, None, :]
        add_mask = add_mask.masked_fill(pad, neg)
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        x = _qs(x + llama_attention(P, cfg, p + "self_attn.", rms_norm(x, P[p + "input_layernorm.weight"], cfg.rms_norm_eps), cos, sin, add_mask))
        h = rms_norm(x, P[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
        h = F.silu(_qo(_linear(h, P[p + "mlp.gate_proj.weight"]))) * _qo(_linear(h, P[p + "mlp.up_proj.weight"]))
        x = _qs(x + _linear(h, P[p + "mlp.down_proj.weight"]))
    return rms_norm(x, P["model.norm.weight"], cfg.rms_norm_eps)


def shifted_ce(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """llama_mmgpt.py:92-100: mean CE over labels[:, 1:] != -100."""
    V = logits.shape[-1]
    return F.cross_entropy(logits[..., :-1, :].reshape(-1, V).float(), labels[..., 1:].reshape(-1), ignore_index=IGNORE_INDEX)


def forward(P: dict, cfg: OracleConfig, input_ids, attention_mask=None, labels=None, images=None):
    """MMGPTLlamaForCausalLM.forward (llama_mmgpt.py:53-112).  Returns (loss|None, logits)."""
    if images is not None and input_ids.shape[1] != 1:
        feats = encode_images(P, cfg, images)
        x = _qs(splice_image_features(P, cfg, input_ids, feats))
    else:
        x = F.embedding(input_ids, P["model.embed_tokens.weight"])
    h = llama_forward(P, cfg, x, attention_mask)
    logits = _linear(h, P["lm_head.weight"])
    loss = shifted_ce(logits, labels) if labels is not None else None
    return loss, logits


def make_params(cfg: OracleConfig, seed: int = 0, requires_grad: bool = False) -> dict:
    """Parameters from the build's counter-based generator (merlin_amd/weights.py)."""
    from merlin_amd import weights as W

    P = {}
    for name, shape in param_shapes(cfg).items():
        t = torch.from_numpy(W.generate(name, shape, seed))
        P[name] = t.requires_grad_(requires_grad)
    return P
