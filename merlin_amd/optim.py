"""Fused AdamW over the flat parameter arena (reference: torch AdamW built by MMGPTTrainer.create_optimizer,
mmgpt/engine/train/trainer.py:45-74: decoupled weight decay, no decay on biases / norm weights).

One `mh_adamw` launch per run of adjacent parameters sharing (lr scale, weight decay): a handful of launches
per layer instead of one per tensor.  fp32 moments live in two flat buffers shaped like the arena."""
from __future__ import annotations

import torch

from . import ops as O


class FusedAdamW:
    def __init__(self, engine, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, lr_scale_fn=None):
        self.engine = engine
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.lr_scale_fn = lr_scale_fn  # name -> multiplier (layer-wise lr decay, llrd_utils.py:4-79)
        self.step_count = 0
        self.m = self.v = None
        self._runs = None
        self._flat_id = None

    def _build(self):
        A = self.engine.ensure_arena()
        if self._flat_id == A.flat.data_ptr() and self._runs is not None:
            return A
        self.m = torch.zeros(A.total, dtype=torch.float32, device=A.flat.device)
        self.v = torch.zeros(A.total, dtype=torch.float32, device=A.flat.device)
        runs = []
        for n in A.names:
            p = A.params[n]
            if not p.requires_grad:
                continue
            wd = 0.0 if (p.dim() <= 1 or n.endswith(".bias")) else self.weight_decay
            sc = 1.0 if self.lr_scale_fn is None else float(self.lr_scale_fn(n))
            off, num = A.offset[n], p.numel()
            if runs and runs[-1][2] == sc and runs[-1][3] == wd and runs[-1][0] + runs[-1][1] <= off and off - (runs[-1][0] + runs[-1][1]) < 256:
                runs[-1][1] = off + num - runs[-1][0]  # merge (alignment gaps hold zeros and stay zero)
            else:
                runs.append([off, num, sc, wd])
        self._runs = runs
        self._flat_id = A.flat.data_ptr()
        return A

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0):
        A = self._build()
        if A.gflat is None:
            return
        self.step_count += 1
        b1, b2 = self.betas
        for off, num, sc, wd in self._runs:
            O.adamw_(A.flat[off: off + num], A.gflat[off: off + num], self.m[off: off + num], self.v[off: off + num],
                     self.lr * sc, b1, b2, self.eps, wd, self.step_count, grad_scale)

    def zero_grad(self, set_to_none=True):
        A = self.engine.arena
        if A is None:
            return
        for p in A.params.values():
            p.grad = None

    def grad_norm(self):
        """Global L2 norm of the gradient arena (device scalar tensor)."""
        A = self.engine.arena
        out = torch.zeros(1, dtype=torch.float32, device=A.flat.device)
        O.sumsq(A.gflat, out)
        return out
