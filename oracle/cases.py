"""Named parity cases (config + synthetic batch) shared by make_golden.py and the tests."""
from __future__ import annotations

import numpy as np

from merlin_amd import synth
from oracle.ref_cpu import OracleConfig

TINY_CASES = ["tiny_1img", "tiny_2img", "tiny_padbatch", "tiny_textonly", "tiny_conv2"]


def tiny_cfg(projector="mlp", conv_stride=1):
    V = 100
    return OracleConfig(vocab_size=V + 3, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                        num_attention_heads=2, rms_norm_eps=1e-6, v_hidden_size=128, v_intermediate_size=256,
                        v_num_hidden_layers=3, v_num_attention_heads=2, v_image_size=56, v_patch_size=14,
                        projector=projector, conv_stride=conv_stride, im_patch_token=V, im_start_token=V + 1,
                        im_end_token=V + 2)


def medium_cfg():
    """Real head dims / widths, 2 layers per tower, S=613 (BASELINE cfg 1 shape)."""
    return OracleConfig(num_hidden_layers=2, v_num_hidden_layers=2)


def full_cfg():
    return OracleConfig()


def released_cfg():
    """Geometry of the released checkpoint (pretrain.sh:6-9, README.md:79-80): CLIP-L/14 at 448 px (32x32 patch grid),
    conv projector stride 2 -> P = 256 image tokens, real widths (1024 / 4096 / 11008, vocab 32003), 2 layers per tower."""
    return OracleConfig(num_hidden_layers=2, v_num_hidden_layers=2, v_image_size=448, projector="conv", conv_stride=2)


def get_case(name):
    if name.startswith("tiny"):
        cfg = tiny_cfg("conv", 2) if name == "tiny_conv2" else tiny_cfg()
        V, P, H = 100, cfg.num_patches, cfg.v_image_size
        T, I = ("tok",), ("img",)
        rng = np.random.RandomState(11)
        if name in ("tiny_1img", "tiny_conv2"):
            return cfg, synth.single_image_batch(V, P, H, n_caption=8, seed=1, img_seed=2)
        if name == "tiny_2img":
            s = synth.pack_sample([("tok", 1, False), ("text", 3, False), I, ("text", 5, True), I, ("tok", 13, False),
                                   ("text", 9, True), ("tok", 2, True)], V, P, rng)
            return cfg, synth.collate([s], H, 21)
        if name == "tiny_padbatch":
            s0 = synth.pack_sample([("tok", 1, False), I, ("text", 6, True), ("tok", 2, True)], V, P, rng)
            s1 = synth.pack_sample([("tok", 1, False), ("text", 2, False), I, ("text", 7, True), I, ("text", 11, True),
                                    ("tok", 2, True)], V, P, rng)
            return cfg, synth.collate([s0, s1], H, 31)
        if name == "tiny_textonly":
            s0 = synth.pack_sample([("tok", 1, False), ("text", 20, True), ("tok", 2, True)], V, P, rng)
            s1 = synth.pack_sample([("tok", 1, False), I, ("text", 12, True), ("tok", 2, True)], V, P, rng)
            return cfg, synth.collate([s0, s1], H, 41)
        raise KeyError(name)
    if name == "medium_cfg1":
        return medium_cfg(), synth.single_image_batch()
    if name == "full_cfg1":
        return full_cfg(), synth.single_image_batch()
    if name == "released_conv448":
        return released_cfg(), synth.single_image_batch(P=256, image_size=448, n_caption=24, seed=5, img_seed=6)
    raise KeyError(name)
