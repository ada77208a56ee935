"""merlin_amd.image_processing against goldens produced by the REAL reference's `BaseDataset.image_processor`
(base_dataset.py:178-197) over transformers' CLIPImageProcessor (oracle/make_image_golden.py): all four aspect-ratio modes,
landscape / portrait / square / upscaled / extreme-aspect inputs.  Bit-exact (same PIL resampling, same float32 arithmetic)."""
import os

import numpy as np
import pytest
from PIL import Image

from merlin_amd import image_processing as IP

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "image_proc.npz"))
S = int(G["image_size"])
N = len([k for k in G.files if k.startswith("in_")])


@pytest.mark.parametrize("mode", ["resize", "pad", "keep", "default"])
@pytest.mark.parametrize("i", range(N))
def test_image_processor_modes_match_reference(i, mode):
    proc = IP.CLIPImageProcessor(size=S)
    out = IP.process_image(Image.fromarray(G[f"in_{i}"]), proc, S, mode).numpy()
    ref = G[f"out_{i}_{mode}"]
    assert out.shape == ref.shape and out.dtype == np.float32
    assert np.abs(out - ref).max() <= 1e-6, np.abs(out - ref).max()


def test_tower_publishes_this_processor():
    """build_vision_tokenizer hands `vision_tower.image_processor` to the data args (base_mmgpt.py:47-51)."""
    from merlin_amd.model import vision

    assert vision.CLIPImageProcessor is IP.CLIPImageProcessor
