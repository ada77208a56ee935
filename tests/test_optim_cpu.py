"""Host logic of the optimizer step against golden vectors captured from the REAL reference
(oracle/make_llrd_golden.py -> tests/golden/llrd_groups.json): layer-wise lr decay grouping (llrd_utils.py:4-79) and
the cosine-with-warmup multipliers HF applies for pretrain.sh's flags."""
import json
import os

import pytest
import torch

from merlin_amd import optim

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "llrd_groups.json")))


class _P:
    def __init__(self, ndim, rg=True):
        self._d, self.requires_grad = ndim, rg

    def dim(self):
        return self._d


def _named(frozen=()):
    return [(n, _P(d, not any(f in n for f in frozen))) for n, d in GOLD["names"]]


@pytest.mark.parametrize("key,fn,frozen", [("vit_llrd", optim.vit_lr_scale, ()), ("llm_llrd", optim.llm_lr_scale, ()), ("plain", None, ()),
                                           ("vit_llrd_frozen_llm", optim.vit_lr_scale, ("model.layers.", "lm_head", "embed_tokens", "model.norm"))])
def test_param_groups_match_reference(key, fn, frozen):
    from oracle import llrd_ref

    got = llrd_ref.param_groups(_named(frozen), GOLD["lr"], GOLD["wd"], fn)
    ref = GOLD[key]
    assert len(got) == len(ref)
    for g, r in zip(got, ref):  # the oracle restatement: same groups, same order, same members, bit-equal lr
        assert g["names"] == r["names"]
        assert g["weight_decay"] == r["weight_decay"]
        assert g["lr"] == r["lr"]
    # the product rule (per-parameter lr / weight decay the fused optimizer applies): same assignment as the reference's groups
    hp = optim.per_param_hparams(_named(frozen), GOLD["lr"], GOLD["wd"], fn)
    want = {n: (r["lr"], r["weight_decay"]) for r in ref for n in r["names"]}
    assert hp == want


def test_lr_scale_values():
    VT = "model.vision_tower.vision_tower.vision_model."
    assert optim.vit_lr_scale(VT + "encoder.layers.22.mlp.fc1.weight") == 1.0
    assert optim.vit_lr_scale(VT + "encoder.layers.0.mlp.fc1.weight") == 0.9 ** 22
    assert optim.vit_lr_scale(VT + "embeddings.class_embedding") == 0.1
    assert optim.vit_lr_scale("model.layers.3.mlp.up_proj.weight") == 1
    assert optim.llm_lr_scale("model.layers.31.mlp.up_proj.weight") == 1.0
    assert optim.llm_lr_scale("model.layers.0.self_attn.q_proj.weight") == 0.931 ** 31
    assert optim.llm_lr_scale("lm_head.weight") == 1


def test_cosine_schedule_matches_hf():
    c = GOLD["cosine"]
    for s, m in zip(c["steps"], c["mult"]):
        assert abs(optim.cosine_with_warmup(s, c["total"], c["warmup_ratio"]) - m) < 1e-12, s
