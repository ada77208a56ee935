"""BASELINE cfg 2 (single 336-px image + 32-token caption, S = 613, B = 1, full-depth ViT-L/14 + Llama-7B) for `rocprofv3 --kernel-trace --stats`:
MODE=fwd (default): 12 forwards under no_grad (= the prefill of every eval `generate`); MODE=train: 8 training steps (fwd + bwd + clip + AdamW)."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa: E402

import bench  # noqa: E402
from merlin_amd import synth  # noqa: E402
from merlin_amd.model.llama_mmgpt import build_synthetic_model  # noqa: E402
from merlin_amd.optim import FusedAdamW, cosine_with_warmup, vit_lr_scale  # noqa: E402

dev = torch.device("cuda", 0)
model = build_synthetic_model(bench.LLAMA_7B, bench.VIT_L_336, projector="mlp", dtype=torch.bfloat16, device=dev, seed=0)
b = synth.single_image_batch()
d = dict(input_ids=b["input_ids"].to(dev), attention_mask=b["attention_mask"].to(dev), labels=b["labels"].to(dev), images=[im.to(dev) for im in b["images"]])
mode = os.environ.get("MODE", "fwd")
if mode == "fwd":
    with torch.no_grad():
        for _ in range(12):
            model(**d)
else:
    model.engine.save_activations = True
    opt = FusedAdamW(model.engine, lr=5e-5, betas=(0.9, 0.95), weight_decay=0.05, lr_scale_fn=vit_lr_scale)
    for i in range(8):
        out = model(**d)
        out.loss.backward()
        opt.step(grad_scale=1.0, max_grad_norm=1.0, lr_mult=cosine_with_warmup(i + 10, 1000, 0.01))
        opt.zero_grad()
torch.cuda.synchronize()
