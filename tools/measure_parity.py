"""Print the measured parity numbers (fp16 / bf16 logits error vs the reference goldens) for the tiny, medium, released and full
cases: the evidence behind the tolerances asserted in tests/.   python tools/measure_parity.py [full]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import cases as C  # noqa: E402
from test_model_gpu import _build, _to_dev  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
names = C.TINY_CASES + ["medium_cfg1", "released_conv448"] + (["full_cfg1"] if "full" in sys.argv else [])
modes = [("fp16", torch.float16, False), ("bf16", torch.bfloat16, False)]
if "parity" in sys.argv:
    modes.append(("fp32-parity", torch.bfloat16, True))
for name in names:
    cfg, batch = C.get_case(name)
    g = np.load(os.path.join(GOLD, name + ".npz"))
    for tag, dt, parity in modes:
        m = _build(cfg, dt)
        if parity:
            m.engine.parity_fp32 = True
        with torch.no_grad():
            out = m(**_to_dev(batch))
        lg = out.logits.float().cpu().numpy()
        if "logits" in g.files:
            mask = batch["attention_mask"].numpy()
            err = np.abs(lg - g["logits"])[mask].max() / np.abs(g["logits"][mask]).max()
        else:
            step = {"medium_cfg1": 8, "released_conv448": 4, "full_cfg1": 16}[name]
            w = {"medium_cfg1": 512, "released_conv448": 512, "full_cfg1": 256}[name]
            err = np.abs(lg[:, ::step, :w] - g["logits_slice"]).max() / float(g["logits_absmax"])
        print(f"{name:18s} {tag:12s} logits max|d|/max|ref| = {err:.3e}   loss {float(out.loss):.6f} ref {float(g['loss']):.6f} rel {abs(float(out.loss) - float(g['loss'])) / float(g['loss']):.2e}", flush=True)
        del m
        torch.cuda.empty_cache()
