"""Parameter-holding module tree with the reference's names (SURVEY.md §8b state-dict keys).

These modules own parameters and expose the attribute surface the reference's builder / trainer
touches (mmgpt/model/builder.py:98-163); they carry NO torch arithmetic: all math runs in the HIP
engine (merlin_amd/model/engine.py).  Linear/Embedding/LayerNorm/Conv2d subclass their torch.nn
counterparts only so `isinstance` checks and state-dict layouts match; their `forward` raises.
"""
from __future__ import annotations

import torch
import torch.nn as nn


def _empty(*shape, device=None, dtype=None):
    return nn.Parameter(torch.empty(*shape, device=device, dtype=dtype))


class _NoTorchMath:
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(f"{type(self).__name__} is a parameter holder; the HIP engine runs the math")


class Linear(_NoTorchMath, nn.Linear):
    def __init__(self, in_features, out_features, bias=True, device=None, dtype=None):
        nn.Module.__init__(self)
        self.in_features, self.out_features = in_features, out_features
        self.weight = _empty(out_features, in_features, device=device, dtype=dtype)
        if bias:
            self.bias = _empty(out_features, device=device, dtype=dtype)
        else:
            self.register_parameter("bias", None)

    def reset_parameters(self):
        pass


class Embedding(_NoTorchMath, nn.Embedding):
    def __init__(self, num_embeddings, embedding_dim, device=None, dtype=None):
        nn.Module.__init__(self)
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.padding_idx = self.max_norm = None
        self.norm_type, self.scale_grad_by_freq, self.sparse = 2.0, False, False
        self.weight = _empty(num_embeddings, embedding_dim, device=device, dtype=dtype)

    def reset_parameters(self):
        pass


class LayerNorm(_NoTorchMath, nn.LayerNorm):
    def __init__(self, dim, eps=1e-5, device=None, dtype=None):
        nn.Module.__init__(self)
        self.normalized_shape, self.eps, self.elementwise_affine = (dim,), eps, True
        self.weight = _empty(dim, device=device, dtype=dtype)
        self.bias = _empty(dim, device=device, dtype=dtype)

    def reset_parameters(self):
        pass


class Conv2d(_NoTorchMath, nn.Module):
    """weight [out, in, kh, kw] (+bias): CLIP patch embedding (no bias) / ConvProjector."""

    def __init__(self, cin, cout, k, stride, padding=0, bias=True, device=None, dtype=None):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding = cin, cout, (k, k), (stride, stride), (padding, padding)
        self.weight = _empty(cout, cin, k, k, device=device, dtype=dtype)
        if bias:
            self.bias = _empty(cout, device=device, dtype=dtype)
        else:
            self.register_parameter("bias", None)


class LlamaRMSNorm(_NoTorchMath, nn.Module):
    def __init__(self, dim, eps, device=None, dtype=None):
        super().__init__()
        self.variance_epsilon = eps
        self.weight = _empty(dim, device=device, dtype=dtype)


# ---- Llama ---------------------------------------------------------------------------------
class LlamaAttention(_NoTorchMath, nn.Module):
    def __init__(self, cfg, **kw):
        super().__init__()
        d = cfg.hidden_size
        self.q_proj, self.k_proj = Linear(d, d, False, **kw), Linear(d, d, False, **kw)
        self.v_proj, self.o_proj = Linear(d, d, False, **kw), Linear(d, d, False, **kw)


class LlamaMLP(_NoTorchMath, nn.Module):
    def __init__(self, cfg, **kw):
        super().__init__()
        d, ff = cfg.hidden_size, cfg.intermediate_size
        self.gate_proj, self.up_proj, self.down_proj = Linear(d, ff, False, **kw), Linear(d, ff, False, **kw), Linear(ff, d, False, **kw)


class LlamaDecoderLayer(_NoTorchMath, nn.Module):
    def __init__(self, cfg, **kw):
        super().__init__()
        self.self_attn = LlamaAttention(cfg, **kw)
        self.mlp = LlamaMLP(cfg, **kw)
        self.input_layernorm = LlamaRMSNorm(cfg.hidden_size, cfg.rms_norm_eps, **kw)
        self.post_attention_layernorm = LlamaRMSNorm(cfg.hidden_size, cfg.rms_norm_eps, **kw)


# ---- CLIP vision (transformers 4.31 nesting: CLIPVisionModel.vision_model.*) ---------------------
class CLIPVisionEmbeddings(_NoTorchMath, nn.Module):
    def __init__(self, vc, **kw):
        super().__init__()
        g = vc.image_size // vc.patch_size
        self.class_embedding = _empty(vc.hidden_size, device=kw.get("device"), dtype=kw.get("dtype"))
        self.patch_embedding = Conv2d(3, vc.hidden_size, vc.patch_size, vc.patch_size, bias=False, **kw)
        self.position_embedding = Embedding(g * g + 1, vc.hidden_size, **kw)


class CLIPAttention(_NoTorchMath, nn.Module):
    def __init__(self, vc, **kw):
        super().__init__()
        d = vc.hidden_size
        self.q_proj, self.k_proj, self.v_proj, self.out_proj = (Linear(d, d, True, **kw) for _ in range(4))


class CLIPMLP(_NoTorchMath, nn.Module):
    def __init__(self, vc, **kw):
        super().__init__()
        self.fc1 = Linear(vc.hidden_size, vc.intermediate_size, True, **kw)
        self.fc2 = Linear(vc.intermediate_size, vc.hidden_size, True, **kw)


class CLIPEncoderLayer(_NoTorchMath, nn.Module):
    def __init__(self, vc, **kw):
        super().__init__()
        self.layer_norm1 = LayerNorm(vc.hidden_size, vc.layer_norm_eps, **kw)
        self.self_attn = CLIPAttention(vc, **kw)
        self.layer_norm2 = LayerNorm(vc.hidden_size, vc.layer_norm_eps, **kw)
        self.mlp = CLIPMLP(vc, **kw)


class CLIPEncoder(_NoTorchMath, nn.Module):
    def __init__(self, vc, **kw):
        super().__init__()
        self.layers = nn.ModuleList([CLIPEncoderLayer(vc, **kw) for _ in range(vc.num_hidden_layers)])


class CLIPVisionTransformer(_NoTorchMath, nn.Module):
    def __init__(self, vc, **kw):
        super().__init__()
        self.embeddings = CLIPVisionEmbeddings(vc, **kw)
        self.pre_layrnorm = LayerNorm(vc.hidden_size, vc.layer_norm_eps, **kw)  # (sic) HF's spelling
        self.encoder = CLIPEncoder(vc, **kw)
        self.post_layernorm = LayerNorm(vc.hidden_size, vc.layer_norm_eps, **kw)


class CLIPVisionModel(_NoTorchMath, nn.Module):
    """Stands in for transformers.CLIPVisionModel (clip_encoder.py:22): `.vision_model.*`, `.config`."""

    def __init__(self, vc, **kw):
        super().__init__()
        self.config = vc
        self.vision_model = CLIPVisionTransformer(vc, **kw)

    @property
    def dtype(self):
        return self.vision_model.pre_layrnorm.weight.dtype

    @property
    def device(self):
        return self.vision_model.pre_layrnorm.weight.device
