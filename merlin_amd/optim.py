"""Fused AdamW over the flat parameter arena (reference: torch AdamW built by MMGPTTrainer.create_optimizer,
mmgpt/engine/train/trainer.py:45-74: decoupled weight decay, no decay on biases / norm weights).

One `mh_adamw` launch per run of adjacent parameters sharing (lr scale, weight decay): a handful of launches
per layer instead of one per tensor.  fp32 moments live in two flat buffers shaped like the arena."""
from __future__ import annotations

import math
import re

import torch

from . import ops as O


# ---- layer-wise learning-rate decay (mmgpt/utils/llrd_utils.py:4-23), selected by --llrd / --llm_llrd (trainer.py:56-61) ----
def vit_lr_scale(name: str) -> float:
    """CLIP tower: 0.9 ** (22 - layer) for encoder layers, 0.1 for the rest of the vision model, 1 elsewhere."""
    if "vision_model.encoder.layers" in name:
        layer = int(re.findall(r"layers\.(\d+)\.", name)[0])
        return 0.9 ** (23 - layer - 1)
    if "vision_model" in name:
        return 0.1
    return 1


def llm_lr_scale(name: str) -> float:
    """Decoder: 0.931 ** (31 - layer) for `model.layers.N.`, 1 elsewhere."""
    if "model.layers" in name:
        layer = int(re.findall(r"layers\.(\d+)\.", name)[0])
        return 0.931 ** (32 - layer - 1)
    return 1


def per_param_hparams(named_parameters, lr, weight_decay, lr_scale_fn=None):
    """{name: (lr, weight_decay)} for the trainable parameters: the two rules of the reference's optimizer construction
    (trainer.py:45-74 / llrd_utils.py:26-79) - biases and 1-D tensors are not decayed; lr is scaled per layer by lr_scale_fn.
    (The reference materialises this as torch param groups; the fused optimizer keeps it per arena range instead.)"""
    out = {}
    for name, prm in named_parameters:
        if not prm.requires_grad:
            continue
        wd = 0.0 if (name.endswith(".bias") or prm.dim() == 1) else weight_decay
        out[name] = (lr * (lr_scale_fn(name) if lr_scale_fn is not None else 1), wd)
    return out


def cosine_with_warmup(step: int, total_steps: int, warmup_ratio: float = 0.0, num_cycles: float = 0.5) -> float:
    """lr multiplier of HF's `cosine` scheduler as the reference launches it (pretrain.sh:28-29: --warmup_ratio 0.01
    --lr_scheduler_type cosine): linear warm-up over ceil(total * ratio) steps, then half a cosine to zero."""
    warmup = math.ceil(total_steps * warmup_ratio)
    if step < warmup:
        return float(step) / float(max(1, warmup))
    progress = float(step - warmup) / float(max(1, total_steps - warmup))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * num_cycles * 2.0 * progress)))


class FusedAdamW:
    def __init__(self, engine, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, lr_scale_fn=None):
        self.engine = engine
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.lr_scale_fn = lr_scale_fn  # name -> multiplier (layer-wise lr decay, llrd_utils.py:4-79)
        self.step_count = 0
        self.m = self.v = None
        self._runs = None
        self._flat_id = None
        self._clip = None
        self._ss = None

    def _build(self):
        A = self.engine.ensure_arena()
        sig = (A.flat.data_ptr(), tuple(p.requires_grad for p in A.params.values()))  # freeze / unfreeze rebuilds the runs
        if self._flat_id == sig and self._runs is not None:
            return A
        if self.m is None or self.m.numel() != A.total or self.m.device != A.flat.device:
            self.m = torch.zeros(A.total, dtype=torch.float32, device=A.flat.device)
            self.v = torch.zeros(A.total, dtype=torch.float32, device=A.flat.device)
        hp = per_param_hparams(A.params.items(), 1.0, self.weight_decay, self.lr_scale_fn)  # lr as a multiplier of self.lr
        runs = []
        prev = -2  # index (arena order) of the last trainable parameter: runs only merge DIRECTLY adjacent parameters
        for idx, n in enumerate(A.names):
            p = A.params[n]
            if not p.requires_grad:
                continue
            sc, wd = hp[n]
            off, num = A.offset[n], p.numel()
            if runs and prev == idx - 1 and runs[-1][2] == sc and runs[-1][3] == wd:
                runs[-1][1] = off + num - runs[-1][0]  # merge (alignment gaps between neighbours hold zeros and stay zero)
            else:
                runs.append([off, num, sc, wd])
            prev = idx
        self._runs = runs
        self._flat_id = sig
        return A

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0, max_grad_norm: float = None, lr_mult: float = 1.0):
        """One AdamW update.  grad_scale folds 1/world_size (and loss scaling) into the kernel; max_grad_norm clips the
        global gradient norm like HF Trainer (--max_grad_norm, default 1.0) with the coefficient computed and applied
        on the device; lr_mult is the scheduler's multiplier for this step (cosine_with_warmup)."""
        A = self._build()
        if A.gflat is None:
            return
        self.step_count += 1
        b1, b2 = self.betas
        clip = None
        if max_grad_norm is not None and max_grad_norm > 0:
            if self._clip is None or self._clip.device != A.flat.device:
                self._clip = torch.zeros(2, dtype=torch.float32, device=A.flat.device)
            O.clip_scale(self.grad_norm_sq(), grad_scale, max_grad_norm, self._clip)
            clip = self._clip
        for off, num, sc, wd in self._runs:
            args = (A.flat[off: off + num], A.gflat[off: off + num], self.m[off: off + num], self.v[off: off + num],
                    self.lr * sc * lr_mult, b1, b2, self.eps, wd, self.step_count, grad_scale)
            if clip is None:
                O.adamw_(*args)
            else:
                O.adamw_clip_(*args, clip)
        self.engine.weights_changed()  # derived copies (fp8 weights, padded patch-embedding weight) are stale now

    def last_grad_norm(self):
        """Total gradient norm seen by the last clipped step (device scalar; what HF logs as grad_norm)."""
        return None if self._clip is None else self._clip[1]

    def zero_grad(self, set_to_none=True):
        A = self.engine.arena
        if A is None:
            return
        for p in A.params.values():
            p.grad = None

    def grad_norm_sq(self):
        """Sum of squares of the gradient arena (device scalar tensor; frozen parameters hold zeros)."""
        A = self.engine.arena
        if self._ss is None or self._ss.device != A.flat.device:
            self._ss = torch.zeros(2049, dtype=torch.float32, device=A.flat.device)
        O.sumsq_det(A.gflat, self._ss[1:], self._ss[:1])  # fixed-order reduction: reproducible clip coefficient
        return self._ss[:1]

    def grad_norm(self):
        """Global L2 norm of the gradient arena (device scalar tensor)."""
        return self.grad_norm_sq().sqrt()
