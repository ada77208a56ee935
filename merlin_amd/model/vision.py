"""Vision tower and projectors with the reference's module surface.

  CLIPVisionTower   <- mmgpt/model/vision_encoder/clip_encoder.py:11-107
  MLPProjector      <- mmgpt/model/projector/mlp_projector.py:11-23
  ConvProjector     <- mmgpt/model/projector/conv_projector.py:8-39
  build_vision_tower / build_projector <- .../vision_encoder/builder.py:7-17, projector/builder.py:8-42

The modules hold parameters (reference state-dict names) and delegate all math to the HIP engine of
the owning MMGPTLlamaForCausalLM; called standalone they run the same engine code paths.
"""
from __future__ import annotations

import json
import os

import torch
import torch.nn as nn

from . import modules as M
from .config import CLIPVisionConfig

from ..image_processing import CLIPImageProcessor  # noqa: E402,F401  (the data path's processor: base_dataset.py:178-197)


def _load_vision_config(name_or_cfg):
    if isinstance(name_or_cfg, CLIPVisionConfig):
        return name_or_cfg
    if isinstance(name_or_cfg, dict):
        return CLIPVisionConfig(**name_or_cfg)
    if isinstance(name_or_cfg, str) and os.path.exists(os.path.join(name_or_cfg, "config.json")):
        return CLIPVisionConfig.from_pretrained(name_or_cfg)
    known = {"vit-large-patch14-336": dict(image_size=336), "vit-large-patch14": dict(image_size=224)}
    for k, v in known.items():
        if isinstance(name_or_cfg, str) and name_or_cfg.rstrip("/").endswith(k):
            return CLIPVisionConfig(**v)
    raise FileNotFoundError(f"no CLIP vision config at {name_or_cfg!r} (expected a directory with config.json)")


class CLIPVisionTower(nn.Module):
    def __init__(self, args, vision_config=None):
        super().__init__()
        self.vision_tower_name = args.vision_tower
        self.select_layer = args.vision_select_layer
        self.select_feature = args.vision_select_feature
        self.freeze_vision_tower = args.freeze_vision_tower
        self.conv_stride = args.conv_stride
        vc = _load_vision_config(vision_config if vision_config is not None else args.vision_tower)
        self.image_processor = CLIPImageProcessor(vc.image_size)
        self.vision_tower = M.CLIPVisionModel(vc)
        self._engine_owner = None  # set by MMGPTLlamaForCausalLM.build_vision_tokenizer
        if self.select_feature not in ("patch", "cls_patch"):
            raise ValueError(f"Unexpected select feature: {self.select_feature}")
        self.load_weights(getattr(args, "model_name_or_path", None))

    def load_weights(self, model_path):
        """clip_encoder.py:26-62: pick `model.vision_tower.*` keys out of the LLM checkpoint, if any."""
        from ..checkpoint import load_prefixed_weights

        if model_path:
            load_prefixed_weights(self, model_path, "model.vision_tower.")

    @property
    def layers_used(self) -> int:
        n = self.config.num_hidden_layers + 1
        return self.select_layer if self.select_layer >= 0 else n + self.select_layer

    def forward(self, images):
        """list[B] of [n_i,3,H,W] -> tuple of [n_i, P, hidden] (dtype of images[0], clip_encoder.py:80)."""
        owner = self._engine_owner() if self._engine_owner is not None else None
        if owner is None:
            raise RuntimeError("CLIPVisionTower runs inside MMGPTLlamaForCausalLM (build_vision_tokenizer attaches the HIP engine)")
        return owner.engine.tower_forward_public(images)

    @property
    def dummy_feature(self):
        return torch.zeros(1, self.hidden_size, device=self.device, dtype=self.dtype)

    @property
    def dtype(self):
        return self.vision_tower.dtype

    @property
    def device(self):
        return self.vision_tower.device

    @property
    def config(self):
        return self.vision_tower.config

    @property
    def hidden_size(self):
        return self.config.hidden_size

    @property
    def num_patches(self):
        return (self.config.image_size // self.config.patch_size // self.conv_stride) ** 2


def build_vision_tower(vision_tower_cfg, vision_config=None):
    name = getattr(vision_tower_cfg, "vision_tower", None)
    if name is None:
        raise ValueError("Unknown vision tower: None")
    # the reference dispatches on substrings ('clip', 'sam', 'qwen', ...); only CLIP is on the hot path
    return CLIPVisionTower(vision_tower_cfg, vision_config=vision_config)


class BaseProjector(nn.Module):
    def load_weights(self, model_path):
        """base_projector.py:12-48: pick `model.projector.*` keys out of the LLM checkpoint, if any."""
        from ..checkpoint import load_prefixed_weights

        if model_path:
            load_prefixed_weights(self, model_path, "model.projector.")

    def forward(self, features):
        owner = self._engine_owner() if getattr(self, "_engine_owner", None) is not None else None
        if owner is None:
            raise RuntimeError("projector runs inside MMGPTLlamaForCausalLM (HIP engine)")
        return owner.engine.projector_forward_public(features)


class MLPProjector(BaseProjector):
    def __init__(self, args, vision_hidden_size, lm_hidden_size):
        super().__init__()
        self.freeze_projector = args.freeze_projector
        self.projector = M.Linear(vision_hidden_size, lm_hidden_size, True)
        self.load_weights(getattr(args, "model_name_or_path", None))


class ConvProjector(BaseProjector):
    def __init__(self, args, vision_hidden_size, lm_hidden_size, conv_stride=1):
        super().__init__()
        self.conv_stride = conv_stride
        self.freeze_projector = args.freeze_projector
        self.projector = M.Conv2d(vision_hidden_size, lm_hidden_size, 3, conv_stride, padding=1, bias=True)
        self.load_weights(getattr(args, "model_name_or_path", None))


def build_projector(projector_cfg, vision_hidden_size, lm_hidden_size):
    kind = getattr(projector_cfg, "projector", None)
    if kind == "mlp":
        return MLPProjector(projector_cfg, vision_hidden_size, lm_hidden_size)
    if kind == "conv":
        return ConvProjector(projector_cfg, vision_hidden_size, lm_hidden_size, conv_stride=projector_cfg.conv_stride)
    raise ValueError(f"Unknown projector: {kind}")
