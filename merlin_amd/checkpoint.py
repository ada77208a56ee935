"""Checkpoint I/O in the reference's layout (SURVEY.md §8b state-dict keys; row N4).

Loads `pytorch_model.bin[.index.json]` / `model.safetensors[.index.json]` shards, selecting keys by
prefix exactly like the reference's tower/projector loaders (clip_encoder.py:26-62,
base_projector.py:12-48); a directory without weights is a silent no-op, as in the reference."""
from __future__ import annotations

import json
import os
from collections import defaultdict

import torch


def _shards(model_path):
    for idx, single in (("pytorch_model.bin.index.json", "pytorch_model.bin"), ("model.safetensors.index.json", "model.safetensors")):
        ip = os.path.join(model_path, idx)
        if os.path.exists(ip):
            wm = json.load(open(ip))["weight_map"]
            by = defaultdict(list)
            for k, f in wm.items():
                by[f].append(k)
            return dict(by)
        sp = os.path.join(model_path, single)
        if os.path.exists(sp):
            return {single: None}
    return {}


def _read(path):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file

        return load_file(path)
    return torch.load(path, map_location="cpu")


def iter_checkpoint(model_path, key_filter=None):
    for fname, keys in _shards(model_path).items():
        if keys is not None and key_filter is not None and not any(key_filter(k) for k in keys):
            continue
        sd = _read(os.path.join(model_path, fname))
        for k, v in sd.items():
            if key_filter is None or key_filter(k):
                yield k, v


def load_prefixed_weights(module, model_path, prefix, strict=True):
    """module.load_state_dict({k[len(prefix):]: v for keys starting with prefix}); no-op if none."""
    if not model_path or not os.path.isdir(model_path):
        return False
    sd = {k[len(prefix):]: v for k, v in iter_checkpoint(model_path, lambda k: k.startswith(prefix))}
    if not sd:
        return False
    module.load_state_dict(sd, strict=strict)
    return True


def save_state_dict(model, path, max_shard_numel=2_500_000_000):
    """Write pytorch_model.bin shards + index with the reference's key names (trainer.py:29-43)."""
    os.makedirs(path, exist_ok=True)
    sd = {k: v.detach().to("cpu") for k, v in model.state_dict().items()}
    shards, cur, n = [], {}, 0
    for k, v in sd.items():
        if n + v.numel() > max_shard_numel and cur:
            shards.append(cur)
            cur, n = {}, 0
        cur[k] = v
        n += v.numel()
    shards.append(cur)
    if len(shards) == 1:
        torch.save(shards[0], os.path.join(path, "pytorch_model.bin"))
        return
    wm = {}
    for i, sh in enumerate(shards):
        fn = f"pytorch_model-{i + 1:05d}-of-{len(shards):05d}.bin"
        torch.save(sh, os.path.join(path, fn))
        wm.update({k: fn for k in sh})
    json.dump({"metadata": {}, "weight_map": wm}, open(os.path.join(path, "pytorch_model.bin.index.json"), "w"))


def interpolate_pos_embed(pos_embed: torch.Tensor, factor: int = 2, extra_tokens: int = 1) -> torch.Tensor:
    """Resize a CLIP position-embedding table [extra + g*g, d] to a (factor*g)^2 grid: bicubic, align_corners=True, the
    class-token row(s) kept (mmgpt/utils/interpolate_model.py:12-30, used to take the 224-px... 336-px tower to the
    released 448-px checkpoint).  Returns the new table; the caller also resets `position_ids` to arange(new_len)."""
    import torch.nn.functional as F

    g = round((pos_embed.shape[0] - extra_tokens) ** 0.5)
    assert g * g + extra_tokens == pos_embed.shape[0], "not a square grid"
    tok, img = pos_embed[:extra_tokens], pos_embed[extra_tokens:]
    img = img.reshape(1, g, g, -1).permute(0, 3, 1, 2)
    img = F.interpolate(img.float(), size=(g * factor, g * factor), mode="bicubic", align_corners=True).to(pos_embed.dtype)
    img = img.permute(0, 2, 3, 1).reshape(g * factor * g * factor, -1)
    return torch.cat([tok, img], dim=0)
