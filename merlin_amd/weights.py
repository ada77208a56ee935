"""Counter-based, platform-independent weight generator.

Every parameter element is a pure function of (state-dict key, element index, seed): a
splitmix64 hash whose four 16-bit fields are summed (Irwin-Hall, n=4) to give an
approximately normal integer, scaled by ONE fp32 multiply and rounded to bf16 (RNE).  Only
integer arithmetic plus one IEEE multiply/add is involved, so the numpy version here and
the HIP kernel `mh_fill_normal` (merlin_amd/csrc/fill.hip) produce identical bits on any
box.  Values with |v| < 2^-14 are flushed to zero so every generated value is exactly
representable in bf16 AND fp16 (and of course fp32): the CPU oracle (fp32) and the HIP
path (bf16 or fp16) therefore start from the same real numbers, and parity measures the
kernels, not weight quantisation.

This replaces `from_pretrained` checkpoints for synthetic runs (no network, SURVEY.md
§8c "Weight generator").  Matrices / embeddings / conv: N(0, 0.02^2).  Norm weights:
1 + N(0, 0.02^2).  Biases: N(0, 0.02^2) (non-zero on purpose: a dropped bias or gamma is
then visible in parity tests).
"""
from __future__ import annotations

import math

import numpy as np

MASK64 = (1 << 64) - 1
_IH_STD = 65536.0 * math.sqrt(1.0 / 3.0)  # std of the sum of four U{0..65535} (sqrt is correctly rounded: same in C)
_IH_MEAN = 131070  # 4 * 65535 / 2


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & MASK64
    return h


def splitmix64_scalar(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & MASK64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def param_key(name: str, seed: int = 0) -> int:
    """64-bit stream key for a parameter (state-dict key + seed)."""
    return splitmix64_scalar(fnv1a64(name) ^ ((seed * 0xD1342543DE82EF95) & MASK64))


def _splitmix64_np(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def round_to_bf16_f32(v: np.ndarray) -> np.ndarray:
    """fp32 -> nearest-even bf16, returned as fp32 (finite inputs only)."""
    u = v.astype(np.float32).view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    u = (u + r) & np.uint32(0xFFFF0000)
    return u.view(np.float32)


def scale_f32(sigma: float) -> np.float32:
    return np.float32(sigma / _IH_STD)


def normal_values(key: int, start: int, count: int, sigma: float = 0.02, offset: float = 0.0) -> np.ndarray:
    """Elements [start, start+count) of stream `key` as fp32 (bf16/fp16-representable)."""
    idx = np.arange(start, start + count, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = _splitmix64_np(idx + np.uint64(key))
    m = np.uint64(0xFFFF)
    s = (z & m) + ((z >> np.uint64(16)) & m) + ((z >> np.uint64(32)) & m) + (z >> np.uint64(48))
    c = (s.astype(np.int64) - _IH_MEAN).astype(np.float32)
    v = c * scale_f32(sigma)
    if offset != 0.0:
        v = v + np.float32(offset)
    v = round_to_bf16_f32(v)
    v[np.abs(v) < np.float32(2.0 ** -14)] = 0.0
    return v


def kind_of(name: str) -> tuple[float, float]:
    """(sigma, offset) by parameter role."""
    leaf = name.rsplit(".", 1)[-1]
    parent = name.rsplit(".", 2)[-2] if name.count(".") >= 1 else ""
    is_norm = ("norm" in parent) or ("layrnorm" in parent)
    if is_norm and leaf == "weight":
        return 0.02, 1.0
    return 0.02, 0.0


def generate(name: str, shape, seed: int = 0, chunk: int = 1 << 24) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    key = param_key(name, seed)
    sigma, offset = kind_of(name)
    out = np.empty(n, dtype=np.float32)
    for s in range(0, n, chunk):
        c = min(chunk, n - s)
        out[s:s + c] = normal_values(key, s, c, sigma, offset)
    return out.reshape(shape)


def fill_state_dict_(named_tensors, seed: int = 0) -> None:
    """In-place fill of {name: torch.Tensor} (CPU) from the generator."""
    import torch

    for name, t in named_tensors.items():
        v = torch.from_numpy(generate(name, tuple(t.shape), seed))
        with torch.no_grad():
            t.copy_(v.to(t.dtype))
