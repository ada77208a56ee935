"""Golden vectors for the optimizer-side host logic, generated from the REAL reference in the build
container: parameter grouping of mmgpt/utils/llrd_utils.py (get_param_groups with vit_lr_scale_func / llm_lr_scale_func /
None) over the full-size parameter name list, and HF's cosine-with-warmup multipliers as pretrain.sh configures them.
Run here only (reads /root/reference); the output tests/golden/llrd_groups.json travels.
usage: python oracle/make_llrd_golden.py"""
import importlib.util
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
spec = importlib.util.spec_from_file_location("ref_llrd", "/root/reference/mmgpt/utils/llrd_utils.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


class P:  # stands in for nn.Parameter: the reference reads .requires_grad and .shape only
    def __init__(self, ndim, rg=True):
        self.shape = (4,) * ndim
        self.requires_grad = rg


def names():
    out = []
    VT = "model.vision_tower.vision_tower.vision_model."
    out += [(VT + "embeddings.class_embedding", 1), (VT + "embeddings.patch_embedding.weight", 4), (VT + "embeddings.position_embedding.weight", 2),
            (VT + "pre_layrnorm.weight", 1), (VT + "pre_layrnorm.bias", 1)]
    for i in range(24):
        p = VT + f"encoder.layers.{i}."
        for m in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.out_proj", "mlp.fc1", "mlp.fc2"):
            out += [(p + m + ".weight", 2), (p + m + ".bias", 1)]
        for m in ("layer_norm1", "layer_norm2"):
            out += [(p + m + ".weight", 1), (p + m + ".bias", 1)]
    out += [(VT + "post_layernorm.weight", 1), (VT + "post_layernorm.bias", 1)]
    out += [("model.projector.projector.weight", 2), ("model.projector.projector.bias", 1), ("model.embed_tokens.weight", 2)]
    for i in range(32):
        p = f"model.layers.{i}."
        for m in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"):
            out.append((p + m + ".weight", 2))
        out += [(p + "input_layernorm.weight", 1), (p + "post_attention_layernorm.weight", 1)]
    out += [("model.norm.weight", 1), ("lm_head.weight", 2)]
    return out


class M:
    def __init__(self, frozen=()):
        self.ps = [(n, P(d, not any(f in n for f in frozen))) for n, d in names()]

    def named_parameters(self):
        return iter(self.ps)


def dump(model, fn, lr, wd):
    groups = ref.get_param_groups(model, None, fn, lr, wd)
    ids = {id(p): n for n, p in model.ps}
    return [{"lr": g["lr"], "weight_decay": g["weight_decay"], "names": [ids[id(p)] for p in g["params"]]} for g in groups]


gold = {"names": names(), "lr": 5e-5, "wd": 0.05,
        "vit_llrd": dump(M(), ref.vit_lr_scale_func, 5e-5, 0.05),
        "llm_llrd": dump(M(), ref.llm_lr_scale_func, 5e-5, 0.05),
        "plain": dump(M(), None, 5e-5, 0.05),
        "vit_llrd_frozen_llm": dump(M(frozen=("model.layers.", "lm_head", "embed_tokens", "model.norm")), ref.vit_lr_scale_func, 5e-5, 0.05)}

from transformers import get_cosine_schedule_with_warmup  # noqa: E402
import math  # noqa: E402

total = 2000
opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
sch = get_cosine_schedule_with_warmup(opt, num_warmup_steps=math.ceil(total * 0.01), num_training_steps=total)
mult = []
for s in range(total):
    mult.append(opt.param_groups[0]["lr"])
    opt.step(); sch.step()
gold["cosine"] = {"total": total, "warmup_ratio": 0.01, "steps": list(range(0, total, 37)) + [total - 1],
                  "mult": [mult[s] for s in list(range(0, total, 37)) + [total - 1]]}
os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
json.dump(gold, open(os.path.join(ROOT, "tests", "golden", "llrd_groups.json"), "w"))
print("groups:", {k: len(v) for k, v in gold.items() if isinstance(v, list) and k != "names"})
