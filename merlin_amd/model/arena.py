"""Flat parameter / gradient arenas.

Every parameter of the model is a VIEW into one flat 16-bit device buffer, laid out layer by
layer with q|k|v and gate|up adjacent, so that
  * the fused QKV and gate+up weights are plain contiguous [3d, d] / [2ff, d] slices (one GEMM),
  * a layer's gradients are one contiguous range = one RCCL all-reduce bucket (merlin_amd/dp.py),
  * the optimizer is one fused multi-tensor AdamW launch per range.
288 GB of HBM makes full replicas (weights + grads + fp32 moments of a 7B model) fit trivially;
nothing is sharded.  State-dict keys are untouched: `state_dict()` sees ordinary parameters.

`nn.Module.to()/.half()` replace `param.data` with fresh tensors and break the views, so the arena
is (re)built lazily: `ensure_packed()` is called at the top of every forward and repacks only when
some parameter no longer points into the arena.
"""
from __future__ import annotations

import torch

ALIGN = 128  # elements


def _round_up(x, m):
    return (x + m - 1) // m * m


class Arena:
    def __init__(self, named_params, alloc_numel=None):
        """named_params: ordered list of (name, Parameter).  alloc_numel: {name: padded numel}."""
        self.names = [n for n, _ in named_params]
        self.params = {n: p for n, p in named_params}
        self.alloc = dict(alloc_numel or {})
        self.offset = {}
        off = 0
        for n, p in named_params:
            self.offset[n] = off
            off += _round_up(max(p.numel(), self.alloc.get(n, 0)), ALIGN)
        self.total = off
        self.flat = None
        self.gflat = None

    # ---- packing -------------------------------------------------------------------------
    def _target(self):
        ps = list(self.params.values())
        dev, dt = ps[0].device, ps[0].dtype
        for p in ps:
            if p.device != dev or p.dtype != dt:
                raise RuntimeError("merlin_amd needs every parameter on one device in one 16-bit dtype; "
                                   f"got {p.device}/{p.dtype} vs {dev}/{dt}")
        return dev, dt

    def is_packed(self) -> bool:
        if self.flat is None:
            return False
        base, es = self.flat.data_ptr(), self.flat.element_size()
        for n, p in self.params.items():
            if p.dtype != self.flat.dtype or p.device != self.flat.device or p.data_ptr() != base + self.offset[n] * es:
                return False
        return True

    def ensure_packed(self) -> bool:
        """Returns True when a repack happened."""
        if self.is_packed():
            return False
        dev, dt = self._target()
        if dev.type != "cuda" or dt not in (torch.bfloat16, torch.float16):
            raise RuntimeError("merlin_amd's HIP path needs bf16/fp16 parameters on a HIP device "
                               f"(got {dt} on {dev}); call model.to(dtype=torch.bfloat16, device='cuda') first")
        flat = torch.zeros(self.total, dtype=dt, device=dev)
        for n, p in self.params.items():
            v = flat[self.offset[n]: self.offset[n] + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
        self.flat = flat
        self.gflat = None
        return True

    def view(self, name, numel=None, shape=None):
        off = self.offset[name]
        n = self.params[name].numel() if numel is None else numel
        v = self.flat[off: off + n]
        if shape is None and numel is None:
            shape = self.params[name].shape
        return v.view(shape) if shape is not None else v

    def span(self, first, last, shape):
        """Contiguous fused view covering params first..last (asserts adjacency, no padding)."""
        i0, i1 = self.names.index(first), self.names.index(last)
        off = self.offset[first]
        n = 0
        for n_ in self.names[i0: i1 + 1]:
            assert self.offset[n_] == off + n, f"{n_} is not adjacent in the arena"
            n += self.params[n_].numel()
        return self.flat[off: off + n].view(shape)

    # ---- gradients ----------------------------------------------------------------------------
    def ensure_grads(self) -> bool:
        """Attach .grad views for trainable params.  Returns `fresh`: True when no trainable parameter
        had a gradient yet (every wgrad of this backward may overwrite instead of accumulate)."""
        if self.gflat is None or self.gflat.dtype != self.flat.dtype or self.gflat.device != self.flat.device:
            self.gflat = torch.zeros_like(self.flat)
            for p in self.params.values():
                p.grad = None
        base, es = self.gflat.data_ptr(), self.gflat.element_size()
        todo = []
        fresh = True
        for n, p in self.params.items():
            if not p.requires_grad:
                continue
            g = p.grad
            if g is not None and g.data_ptr() == base + self.offset[n] * es:
                fresh = False  # already attached and holding accumulated gradients
            else:
                if g is not None:
                    fresh = False  # a foreign grad tensor (set by the user): folded in below
                todo.append((n, p, g))
        for n, p, g in todo:
            v = self.gflat[self.offset[n]: self.offset[n] + p.numel()].view(p.shape)
            if g is not None:
                v.copy_(g)
            elif not fresh:
                v.zero_()
            p.grad = v
        return fresh

    def gview(self, name, shape=None):
        off = self.offset[name]
        v = self.gflat[off: off + self.params[name].numel()]
        return v.view(shape if shape is not None else self.params[name].shape)

    def gspan(self, first, last, shape):
        i0, i1 = self.names.index(first), self.names.index(last)
        off = self.offset[first]
        n = sum(self.params[x].numel() for x in self.names[i0: i1 + 1])
        return self.gflat[off: off + n].view(shape)

    def range_of(self, names):
        """(offset, numel incl. padding) covering a list of adjacent params (for bucketing)."""
        i0 = min(self.names.index(n) for n in names)
        i1 = max(self.names.index(n) for n in names)
        start = self.offset[self.names[i0]]
        end = self.offset[self.names[i1 + 1]] if i1 + 1 < len(self.names) else self.total
        return start, end - start
