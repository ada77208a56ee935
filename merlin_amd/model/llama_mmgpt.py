"""MMGPTLlamaForCausalLM on the MI355X HIP engine: the reference's model surface, our kernels.

Mirrors (file:line under /root/reference):
  mmgpt/model/mmgpt/llama_mmgpt.py:27-35    MMGPTConfig / MMGPTLlamaModel
  mmgpt/model/mmgpt/llama_mmgpt.py:38-134   MMGPTLlamaForCausalLM (forward, prepare_inputs_for_generation)
  mmgpt/model/mmgpt/base_mmgpt.py:13-165    MMGPTMetaForCausalLM (encode_images, build_vision_tokenizer,
                                            prepare_inputs_labels_for_multimodal)
Same constructor, forward signature, attribute names, state-dict keys and errors; the arithmetic is
merlin_amd/model/engine.py (hand-written HIP kernels), and `loss.backward()` runs the engine's manual
backward through a single autograd node that writes `param.grad` views of the gradient arena.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Any, List, Optional

import torch
import torch.nn as nn

from . import modules as M
from .config import MMGPTConfig
from .config import head_dim_of, rope_theta_of  # noqa: F401  (re-exported)
from .engine import HipEngine
from .vision import build_projector, build_vision_tower

IGNORE_INDEX = -100
DEFAULT_IM_PATCH_TOKEN = "<im_patch>"  # mmgpt/utils/constants.py:10-12
DEFAULT_IM_START_TOKEN = "<im_start>"
DEFAULT_IM_END_TOKEN = "<im_end>"


@dataclass
class CausalLMOutputWithPast:
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_key_values: Any = None
    hidden_states: Any = None
    attentions: Any = None

    def __getitem__(self, i):
        vals = tuple(v for v in (self.loss, self.logits, self.past_key_values, self.hidden_states, self.attentions) if v is not None)
        return vals[i]

    def to_tuple(self):
        return tuple(v for v in (self.loss, self.logits, self.past_key_values, self.hidden_states, self.attentions) if v is not None)


def _check_config(config):
    """Accepts merlin_amd's own MMGPTConfig or a transformers LlamaConfig subclass (merlin_amd/hf_compat.py): the engine reads
    vocab_size, hidden_size, intermediate_size, num_hidden_layers, num_attention_heads, rms_norm_eps, rope theta, head_dim."""
    kv = getattr(config, "num_key_value_heads", None)
    if kv not in (None, config.num_attention_heads):
        raise NotImplementedError("GQA is not part of the reference's Llama-7B path")
    if getattr(config, "hidden_act", "silu") != "silu":
        raise NotImplementedError("LlamaMLP uses silu")
    if config.hidden_size % config.num_attention_heads:
        raise ValueError("hidden_size must be a multiple of num_attention_heads")


class MMGPTLlamaModel(nn.Module):
    """LlamaModel parameter tree (`embed_tokens`, `layers`, `norm`) + vision_tower/projector slots."""

    config_class = MMGPTConfig

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.padding_idx = config.pad_token_id
        self.vocab_size = config.vocab_size
        self.embed_tokens = M.Embedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([M.LlamaDecoderLayer(config) for _ in range(config.num_hidden_layers)])
        self.norm = M.LlamaRMSNorm(config.hidden_size, config.rms_norm_eps)


class _HipStep(torch.autograd.Function):
    """One autograd node for the whole model: forward ran in the engine; backward runs the engine's
    hand-written backward and deposits gradients in the arena (returns no tensor gradients)."""

    @staticmethod
    def forward(ctx, anchor, loss, engine, ectx):
        ctx.engine, ctx.ectx = engine, ectx
        return loss.clone()

    @staticmethod
    def backward(ctx, gloss):
        engine, ectx = ctx.engine, ctx.ectx
        ctx.engine = ctx.ectx = None
        if ectx.get("done"):
            raise RuntimeError("merlin_amd: backward through the same forward twice (activations were released)")
        ectx["done"] = True
        engine.backward(ectx, gscale=float(gloss))
        return None, None, None, None


class MMGPTLlamaForCausalLM(nn.Module):
    config_class = MMGPTConfig

    def __init__(self, config):
        super().__init__()
        _check_config(config)
        self.config = config
        self.model = MMGPTLlamaModel(config)
        self.vocab_size = config.vocab_size
        self.lm_head = M.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.engine = HipEngine(self)
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.engine.weights_changed())
        self._anchor = None
        self.use_im_start_end = True
        self.use_beam_search = False
        self.im_patch_token = self.im_start_token = self.im_end_token = None

    # ---- reference surface ---------------------------------------------------------------------
    def get_model(self):
        return self.model

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def get_output_embeddings(self):
        return self.lm_head

    def enable_input_require_grads(self):
        """builder.py:102-108 hooks the embedding output for gradient checkpointing; the engine
        recomputes layers itself, so nothing is needed."""

    def gradient_checkpointing_enable(self, *a, **k):
        self.engine.save_activations = False

    def gradient_checkpointing_disable(self):
        self.engine.save_activations = True

    @property
    def dtype(self):
        return self.lm_head.weight.dtype

    @property
    def device(self):
        return self.lm_head.weight.device

    def resize_token_embeddings(self, new_num_tokens: Optional[int] = None):
        """Grow (or shrink) embed_tokens / lm_head rows, keeping existing rows (HF semantics; new rows
        are zero until the caller initialises them, as build_vision_tokenizer does)."""
        if new_num_tokens is None or new_num_tokens == self.config.vocab_size:
            return self.model.embed_tokens
        d = self.config.hidden_size
        for holder, attr in ((self.model.embed_tokens, "weight"), (self.lm_head, "weight")):
            old = getattr(holder, attr)
            new = torch.zeros(new_num_tokens, d, dtype=old.dtype, device=old.device)
            n = min(old.shape[0], new_num_tokens)
            new[:n].copy_(old.data[:n])
            p = nn.Parameter(new, requires_grad=old.requires_grad)
            setattr(holder, attr, p)
        self.model.embed_tokens.num_embeddings = new_num_tokens
        self.lm_head.out_features = new_num_tokens
        self.config.vocab_size = self.vocab_size = self.model.vocab_size = new_num_tokens
        return self.model.embed_tokens

    def encode_images(self, images):
        """base_mmgpt.py:18-21."""
        feats = self.get_model().vision_tower(images)
        return self.get_model().projector(feats)

    def build_vision_tokenizer(self, model_args, data_args, training_args, tokenizer, vision_config=None):
        """base_mmgpt.py:23-79: build tower+projector, add <im_patch>/<im_start>/<im_end>, resize the
        embeddings and mean-initialise the new rows, publish the data_args the packers read."""
        import weakref

        vision_tower = build_vision_tower(model_args, vision_config=vision_config)
        projector = build_projector(model_args, vision_tower.hidden_size, self.config.hidden_size)
        ref = weakref.ref(self)
        vision_tower._engine_owner = ref
        projector._engine_owner = ref
        ref_p = self.lm_head.weight
        vision_tower.to(device=ref_p.device, dtype=ref_p.dtype)
        projector.to(device=ref_p.device, dtype=ref_p.dtype)
        self.get_model().vision_tower = vision_tower
        self.get_model().projector = projector

        data_args.image_token_len = vision_tower.num_patches
        data_args.image_processor = vision_tower.image_processor
        data_args.use_im_start_end = model_args.use_im_start_end
        self.use_im_start_end = model_args.use_im_start_end
        self.use_beam_search = getattr(data_args, "use_beam_search", False)

        tokenizer.add_tokens([DEFAULT_IM_PATCH_TOKEN], special_tokens=True)
        self.resize_token_embeddings(len(tokenizer))
        self.im_patch_token = tokenizer.convert_tokens_to_ids([DEFAULT_IM_PATCH_TOKEN])[0]
        if self.use_im_start_end:
            self.num_new_tokens = tokenizer.add_tokens([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN], special_tokens=True)
            self.resize_token_embeddings(len(tokenizer))
            self.im_start_token, self.im_end_token = tokenizer.convert_tokens_to_ids([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN])
            if self.num_new_tokens > 0:
                with torch.no_grad():
                    for w in (self.get_input_embeddings().weight, self.get_output_embeddings().weight):
                        avg = w.data[:-self.num_new_tokens].float().mean(dim=0, keepdim=True).to(w.dtype)
                        w.data[-self.num_new_tokens:] = avg

    def set_image_tokens(self, im_patch, im_start, im_end):
        """For callers that configure token ids without a tokenizer (synthetic runs)."""
        self.im_patch_token, self.im_start_token, self.im_end_token = im_patch, im_start, im_end

    # ---- forward ---------------------------------------------------------------------------------
    def forward(self, input_ids=None, attention_mask=None, past_key_values=None, inputs_embeds=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, images=None, return_dict=None):
        if output_attentions or output_hidden_states:
            raise AssertionError("output_attentions / output_hidden_states are not supported (flash path, "
                                 "llama_flash_attn_monkey_patch.py:61)")
        if past_key_values is not None:
            raise AssertionError("past_key_value is not supported (llama_flash_attn_monkey_patch.py:54)")
        return_dict = return_dict if return_dict is not None else self.config.use_return_dict
        if not self.use_im_start_end and images is not None:
            raise NotImplementedError  # base_mmgpt.py:137
        beam_rep = 0
        if self.use_beam_search and images is not None and input_ids is not None and input_ids.shape[1] != 1:
            # base_mmgpt.py:121,160-163: under use_beam_search HF has already expanded input_ids x num_beams while `images` still
            # has one entry per prompt; the reference's zip() keeps the first len(images) rows and repeats the spliced
            # embeddings 5x.  Identical rows give identical logits: compute the kept rows once and repeat the result.
            beam_rep = 5
            n_keep = len(images)
            input_ids = input_ids[:n_keep]
            attention_mask = attention_mask[:n_keep] if attention_mask is not None else None
            labels = labels[:n_keep] if labels is not None else None
        want_grad = torch.is_grad_enabled() and labels is not None and any(p.requires_grad for p in self.parameters())
        # opt-in fp8 paths (BASELINE cfg 5's fp8 MFMA weight path): model.fp8_training = True -> forward + backward GEMMs of the decoder
        # on the scaled-fp8 MFMA; model.fp8_forward = True -> the inference form (forward only, weights quantised once)
        fp8 = "train" if getattr(self, "fp8_training", False) else bool(getattr(self, "fp8_forward", False))
        if getattr(self.engine, "parity_fp32", False) and inputs_embeds is None:
            # fp32-store parity forward (merlin_amd/parity.py): measures the kernels against BASELINE's 1e-3, forward only
            if want_grad:
                raise RuntimeError("engine.parity_fp32 is a forward-only checking mode: run it under torch.no_grad()")
            from .. import parity as _parity

            loss, logits = _parity.forward(self.engine, input_ids, attention_mask, labels, images)
            if beam_rep:
                logits = logits.repeat_interleave(beam_rep, dim=0)
            return CausalLMOutputWithPast(loss=loss, logits=logits) if return_dict else ((loss, logits) if loss is not None else (logits,))
        with torch.no_grad():
            loss, logits, ectx = self.engine.forward(input_ids, attention_mask, labels, images, inputs_embeds=inputs_embeds,
                                                     want_grad=want_grad, fp8=fp8)
        if want_grad:
            if self._anchor is None or self._anchor.device != loss.device:
                self._anchor = torch.zeros(1, device=loss.device, requires_grad=True)
            loss = _HipStep.apply(self._anchor, loss, self.engine, ectx)
        if beam_rep:
            logits = logits.repeat_interleave(beam_rep, dim=0)
        if not return_dict:
            return ((loss, logits) if loss is not None else (logits,))
        return CausalLMOutputWithPast(loss=loss, logits=logits)

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None, **kwargs):
        """llama_mmgpt.py:114-134."""
        if past_key_values:
            input_ids = input_ids[:, -1:]
        if inputs_embeds is not None and past_key_values is None:
            model_inputs = {"inputs_embeds": inputs_embeds}
        else:
            model_inputs = {"input_ids": input_ids}
        model_inputs.update({"past_key_values": past_key_values, "use_cache": kwargs.get("use_cache"),
                             "attention_mask": attention_mask, "images": kwargs.get("images", None)})
        return model_inputs

    def generate(self, input_ids, images=None, **kwargs):
        """HF `generate` as the reference's eval scripts call it (eval_mmvet.py:101-120): greedy, `do_sample=True,
        temperature=...` (HF warpers: temperature, top_k=50 default, top_p) and `num_beams=5` beam search, with
        `stopping_criteria`, `max_new_tokens` / `max_length`, eos / pad handling of transformers' GenerationMixin; runs on the
        engine's prefill + KV-cache decode step (merlin_amd/generation.py).  Extras: use_cache=False (full recompute per token,
        the cross-check), use_graph (decode step as one HIP graph), fp8_weights (fp8 weight copies for the decode GEMVs),
        seed (counter-based sampling stream; default drawn from torch's global generator)."""
        from ..generation import generate as _generate

        return _generate(self, input_ids, images=images, **kwargs)

    # ---- construction helpers ----------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, path, config=None, cache_dir=None, torch_dtype=None, dtype=None, **kw):
        """builder.py:70-74.  Loads config.json and, when present, weights in the reference layout."""
        from ..checkpoint import iter_checkpoint

        config = config if config is not None else cls.config_class.from_pretrained(path)
        model = cls(config)
        own = dict(model.named_parameters())
        for k, v in iter_checkpoint(path, lambda k: k in own) if os.path.isdir(path) else ():
            with torch.no_grad():
                own[k].copy_(v)
        torch_dtype = dtype if dtype is not None else torch_dtype  # (transformers >= 5 spells it `dtype`)
        if torch_dtype is not None:
            model.to(dtype=torch_dtype)
        return model

    def save_pretrained(self, path):
        from ..checkpoint import save_state_dict

        self.config.save_pretrained(path)
        save_state_dict(self, path)

    @torch.no_grad()
    def init_weights_from_generator(self, seed: int = 0):
        """Fill every parameter from the counter-based generator (merlin_amd/weights.py) ON DEVICE with
        mh_fill_normal: synthetic random-init weights, bit-identical to the oracle's numpy stream."""
        from .. import ops as O
        from .. import weights as W

        for name, p in self.named_parameters():
            sigma, offset = W.kind_of(name)
            if p.device.type == "cuda":
                tmp = p.data if p.data.is_contiguous() else torch.empty_like(p.data)
                O.fill_normal_(tmp, W.param_key(name, seed), 0, sigma, offset)
                if tmp is not p.data:
                    p.data.copy_(tmp)
            else:
                p.data.copy_(torch.from_numpy(W.generate(name, tuple(p.shape), seed)).to(p.dtype))
        self.engine.weights_changed()
        return self


class _SynthTokenizer:
    """Minimal tokenizer for synthetic runs: integer vocab + add_tokens (what build_vision_tokenizer needs)."""

    def __init__(self, n):
        self.n, self.names = n, {}

    def add_tokens(self, toks, special_tokens=True):
        k = 0
        for t in toks:
            if t not in self.names:
                self.names[t] = self.n
                self.n += 1
                k += 1
        return k

    def __len__(self):
        return self.n

    def convert_tokens_to_ids(self, toks):
        return [self.names[t] for t in toks]


def build_synthetic_model(llama_cfg: dict, vision_cfg: dict, projector="mlp", conv_stride=1, dtype=torch.bfloat16,
                          device="cuda", seed=0, freeze_vision_tower=False, select_layer=-2, select_feature="patch"):
    """Random-init model in the reference's construction order (builder.py:70-163) without a checkpoint:
    base-vocab Llama -> build_vision_tokenizer (+3 tokens) -> .to(dtype, device) -> generator weights."""
    import types

    from .config import CLIPVisionConfig

    cfg = MMGPTConfig(**llama_cfg)
    old_dtype = torch.get_default_dtype()
    torch.set_default_dtype(dtype)  # create parameters directly in the target dtype on the target device
    try:
        with torch.device(device):
            model = MMGPTLlamaForCausalLM(cfg)
            model = _finish_synthetic(model, cfg, vision_cfg, projector, conv_stride, device, freeze_vision_tower, select_layer, select_feature)
    finally:
        torch.set_default_dtype(old_dtype)
    model.to(dtype=dtype, device=device)
    model.init_weights_from_generator(seed)
    return model


def _finish_synthetic(model, cfg, vision_cfg, projector, conv_stride, device, freeze_vision_tower, select_layer, select_feature):
    import types

    from .config import CLIPVisionConfig

    margs = types.SimpleNamespace(vision_tower="synthetic-clip", vision_select_layer=select_layer, vision_select_feature=select_feature,
                                  freeze_vision_tower=freeze_vision_tower, conv_stride=conv_stride, model_name_or_path=None,
                                  projector=projector, freeze_projector=False, use_im_start_end=True, freeze_lm_model=False)
    dargs = types.SimpleNamespace(use_beam_search=False)
    targs = types.SimpleNamespace(device=device)
    tok = _SynthTokenizer(cfg.vocab_size)
    model.build_vision_tokenizer(margs, dargs, targs, tok, vision_config=CLIPVisionConfig(**vision_cfg))
    model.data_args = dargs
    return model
