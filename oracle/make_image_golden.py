"""Golden vectors for the image preprocessing: the REAL reference's `BaseDataset.image_processor`
(base_dataset.py:178-197, unbound method on a stand-in `self`) driving transformers' CLIPImageProcessor on synthetic PIL images,
all four aspect-ratio modes.  Build container only (reads /root/reference).  -> tests/golden/image_proc.npz (inputs + outputs)."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from PIL import Image  # noqa: E402
from transformers import CLIPImageProcessor  # noqa: E402  (resolved BEFORE the torchvision stub exists: the PIL backend)

proc = CLIPImageProcessor(size={"shortest_edge": 28}, crop_size={"height": 28, "width": 28})
import oracle.make_packer_golden as MP  # noqa: E402,F401  (installs the import stubs and /root/reference on sys.path)

from mmgpt.data.dataset.base_dataset import BaseDataset  # noqa: E402

S = 28
rng = np.random.RandomState(0)
rec = {"image_size": np.int64(S)}
shapes = [(61, 97), (97, 61), (40, 40), (20, 33), (150, 31)]  # (H, W)
for i, (h, w) in enumerate(shapes):
    # smooth-ish content so that bicubic resampling is exercised on non-trivial data
    base = rng.randint(0, 256, size=(h // 4 + 2, w // 4 + 2, 3)).astype(np.uint8)
    img = Image.fromarray(base).resize((w, h), resample=Image.BILINEAR)
    arr = np.asarray(img).copy()
    arr[::7, ::5] = rng.randint(0, 256, size=arr[::7, ::5].shape)
    rec[f"in_{i}"] = arr
    for mode in ("resize", "pad", "keep", "default"):
        me = types.SimpleNamespace(multimodal_cfg={"image_aspect_ratio": mode}, image_size=S, processor=proc)
        out = BaseDataset.image_processor(me, Image.fromarray(arr))
        rec[f"out_{i}_{mode}"] = np.asarray(out, dtype=np.float32)
        print(i, (h, w), mode, tuple(out.shape))
path = os.path.join(ROOT, "tests", "golden", "image_proc.npz")
np.savez_compressed(path, **rec)
print("wrote", path, os.path.getsize(path))
