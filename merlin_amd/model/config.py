"""Configs mirroring the reference's: `MMGPTConfig(LlamaConfig)` with model_type "mmgpt"
(mmgpt/model/mmgpt/llama_mmgpt.py:27-28) and transformers' CLIPVisionConfig field names.

Plain Python (no transformers import on the hot path); `merlin_amd.hf_compat` registers the
AutoConfig/AutoModel entries when transformers is wanted."""
from __future__ import annotations

import copy
import json
import os


class _Cfg:
    def to_dict(self):
        return {k: copy.deepcopy(v) for k, v in self.__dict__.items() if not k.startswith("_")}

    def __repr__(self):
        return f"{type(self).__name__} {json.dumps(self.to_dict(), indent=2, sort_keys=True, default=str)}"


class CLIPVisionConfig(_Cfg):
    model_type = "clip_vision_model"

    def __init__(self, hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                 image_size=336, patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5, num_channels=3, **kw):
        self.hidden_size, self.intermediate_size = hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.image_size, self.patch_size, self.hidden_act = image_size, patch_size, hidden_act
        self.layer_norm_eps, self.num_channels = layer_norm_eps, num_channels
        if hidden_act != "quick_gelu":
            raise NotImplementedError("only CLIP's quick_gelu is implemented")
        for k, v in kw.items():
            setattr(self, k, v)

    @classmethod
    def from_pretrained(cls, path):
        with open(os.path.join(path, "config.json")) as f:
            d = json.load(f)
        d = d.get("vision_config", d)
        return cls(**{k: v for k, v in d.items() if k != "model_type"})


class MMGPTConfig(_Cfg):
    """LlamaConfig fields (transformers) + model_type 'mmgpt'."""

    model_type = "mmgpt"

    def __init__(self, vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                 num_attention_heads=32, num_key_value_heads=None, hidden_act="silu", max_position_embeddings=4096,
                 rms_norm_eps=1e-6, rope_theta=10000.0, use_cache=True, pad_token_id=None, bos_token_id=1,
                 eos_token_id=2, tie_word_embeddings=False, torch_dtype=None, **kw):
        self.vocab_size, self.hidden_size, self.intermediate_size = vocab_size, hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.num_key_value_heads = num_key_value_heads or num_attention_heads
        self.hidden_act, self.max_position_embeddings = hidden_act, max_position_embeddings
        self.rms_norm_eps, self.rope_theta, self.use_cache = rms_norm_eps, rope_theta, use_cache
        self.pad_token_id, self.bos_token_id, self.eos_token_id = pad_token_id, bos_token_id, eos_token_id
        self.tie_word_embeddings, self.torch_dtype = tie_word_embeddings, torch_dtype
        self.output_attentions = kw.pop("output_attentions", False)
        self.output_hidden_states = kw.pop("output_hidden_states", False)
        self.use_return_dict = kw.pop("use_return_dict", True)
        if self.num_key_value_heads != num_attention_heads:
            raise NotImplementedError("GQA is not part of the reference's Llama-7B path")
        if hidden_act != "silu":
            raise NotImplementedError("LlamaMLP uses silu")
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads

    @classmethod
    def from_pretrained(cls, path, **kw):
        with open(os.path.join(path, "config.json")) as f:
            d = json.load(f)
        d.update(kw)
        d.pop("model_type", None)
        d.pop("architectures", None)
        return cls(**d)

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        d = self.to_dict()
        d["model_type"] = self.model_type
        d["architectures"] = ["MMGPTLlamaForCausalLM"]
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(d, f, indent=2, default=str)


def head_dim_of(config) -> int:
    """transformers < 4.45 (the reference pins 4.31.0) has no `LlamaConfig.head_dim`, later versions may carry None."""
    return int(getattr(config, "head_dim", None) or config.hidden_size // config.num_attention_heads)


def rope_theta_of(config) -> float:
    t = getattr(config, "rope_theta", None)
    if t is None:  # transformers >= 5: rope_parameters = {"rope_theta": ..., "rope_type": "default"}
        rp = getattr(config, "rope_parameters", None) or {}
        if rp.get("rope_type", "default") != "default":
            raise NotImplementedError("only the default rotary embedding is on the reference's Llama path")
        t = rp.get("rope_theta", 10000.0)
    return float(t)
