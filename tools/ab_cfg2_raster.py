"""Round 6: the GEMM raster group (tile rows visited together by one XCD, `mh_gemm_raster_group`) at the row counts of a short prefill.
BASELINE cfg 2 is ONE 613-token sequence: 3 tile rows of 256 (5 of 128 in the half-tile forms), and the default group of 4 rows puts the 5th
half-tile row into a group of its own - its tiles re-read every weight panel on other XCDs.  Times the full-depth cfg-2 forward and training
step (the driver's `extras.cfg2` protocol) per group size, alternating, same process.
    python tools/ab_cfg2_raster.py [gm ...]            (on the GPU box; default 4 5 8 3)"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from merlin_amd import ops as O  # noqa: E402
from merlin_amd import synth  # noqa: E402
from merlin_amd.model.llama_mmgpt import build_synthetic_model  # noqa: E402
from merlin_amd.optim import FusedAdamW, cosine_with_warmup, vit_lr_scale  # noqa: E402

dev = torch.device("cuda", 0)
model = build_synthetic_model(bench.LLAMA_7B, bench.VIT_L_336, projector="mlp", dtype=torch.bfloat16, device=dev, seed=0)
model.engine.save_activations = True
b = synth.single_image_batch()
d = dict(input_ids=b["input_ids"].to(dev), attention_mask=b["attention_mask"].to(dev), labels=b["labels"].to(dev), images=[im.to(dev) for im in b["images"]])
opt = FusedAdamW(model.engine, lr=5e-5, betas=(0.9, 0.95), weight_decay=0.05, lr_scale_fn=vit_lr_scale)
it = [0]


def train_step():
    out = model(**d)
    out.loss.backward()
    opt.step(grad_scale=1.0, max_grad_norm=1.0, lr_mult=cosine_with_warmup(it[0] + 10, 1000, 0.01))
    opt.zero_grad()
    it[0] += 1


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def fwd():
    with torch.no_grad():
        model(**d)


gms = [int(a) for a in sys.argv[1:]] or [4, 5, 8, 3]
for _ in range(3):
    train_step()
res = {g: ([], []) for g in gms}
for rep in range(3):
    for g in gms:
        O.gemm_raster_group(g)
        res[g][0].append(timed(fwd, 10))
        res[g][1].append(timed(train_step, 6))
O.gemm_raster_group(4)
for g in gms:
    f, t = res[g]
    print(f"raster group {g:2d}: cfg-2 forward {min(f):7.3f} ms (runs {' '.join(f'{x:.3f}' for x in f)})   training step {min(t):7.2f} ms (runs {' '.join(f'{x:.2f}' for x in t)})", flush=True)
