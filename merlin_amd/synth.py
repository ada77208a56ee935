"""Synthetic batches in the reference collator's layout (mmgpt/data/collator.py:29-34):
    {input_ids int64[B,S], labels int64[B,S] (-100 = ignore), attention_mask bool[B,S],
     images: list[B] of float32[n_i, 3, H, W]}
Sequences follow the reference packers' shape: every image is
<im_start> + P x <im_patch> + <im_end>; image positions and prompts are masked in labels;
N image-text pairs are concatenated into ONE causal sequence with no per-pair mask
(interpair_webdataset.py:63-162).  Token ids ~ U[3, base_vocab), numpy RandomState (frozen
legacy stream => same bits on any box).
"""
from __future__ import annotations

import numpy as np
import torch

IGNORE_INDEX = -100
BOS, EOS, NEWLINE, PAD = 1, 2, 13, 0


def _image_span(P, base_vocab):
    return [base_vocab + 1] + [base_vocab] * P + [base_vocab + 2]


def pack_sample(segments, base_vocab, P, rng):
    """segments: list of ('img',) | ('text', n, supervised: bool) | ('tok', id, supervised)."""
    ids, lab, n_img = [], [], 0
    for seg in segments:
        if seg[0] == "img":
            span = _image_span(P, base_vocab)
            ids += span
            lab += [IGNORE_INDEX] * len(span)
            n_img += 1
        elif seg[0] == "text":
            t = rng.randint(3, base_vocab, size=seg[1]).tolist()
            ids += t
            lab += t if seg[2] else [IGNORE_INDEX] * len(t)
        else:
            ids.append(seg[1])
            lab.append(seg[1] if seg[2] else IGNORE_INDEX)
    return ids, lab, n_img


def collate(samples, image_size, img_seed, zero_image_for_text_only=True):
    """Right-pad like the reference collator; attention_mask = ids != pad."""
    S = max(len(s[0]) for s in samples)
    B = len(samples)
    ids = np.full((B, S), PAD, dtype=np.int64)
    lab = np.full((B, S), IGNORE_INDEX, dtype=np.int64)
    mask = np.zeros((B, S), dtype=bool)
    images = []
    for b, (i, l, n) in enumerate(samples):
        ids[b, : len(i)] = i
        lab[b, : len(l)] = l
        mask[b, : len(i)] = True
        rng = np.random.RandomState(img_seed + b)
        if n == 0:  # interpair_webdataset.py:154-156: text-only samples carry a zeros image
            images.append(torch.zeros(1, 3, image_size, image_size))
        else:
            images.append(torch.from_numpy(rng.standard_normal((n, 3, image_size, image_size)).astype(np.float32)))
    return {"input_ids": torch.from_numpy(ids), "labels": torch.from_numpy(lab),
            "attention_mask": torch.from_numpy(mask), "images": images}


def single_image_batch(base_vocab=32000, P=576, image_size=336, n_caption=32, seed=1, img_seed=2):
    """BASELINE cfg 1/2: [BOS, <im_start>, P x <im_patch>, <im_end>, '\\n', caption, EOS]."""
    rng = np.random.RandomState(seed)
    s = pack_sample([("tok", BOS, False), ("img",), ("tok", NEWLINE, False), ("text", n_caption, True), ("tok", EOS, True)],
                    base_vocab, P, rng)
    return collate([s], image_size, img_seed)


def interpair_batch(B=8, S=4096, frames=6, base_vocab=32000, P=576, image_size=336, rank=0, lead_text=4, ragged=False):
    """BASELINE cfg 3/4: BOS + frames x (lead_text ids + image span) + trajectory ids + EOS = S."""
    samples = []
    for b in range(B):
        rng = np.random.RandomState(100 + b + 1000 * rank)
        fixed = 1 + frames * (lead_text + P + 2) + 1
        tail = S - fixed
        if ragged:
            tail = max(1, tail - (b * 37) % max(1, min(tail - 1, 200)))
        assert tail >= 1, "sequence too short for the requested frames"
        seg = [("tok", BOS, False)]
        for _ in range(frames):
            seg += [("text", lead_text, False), ("img",)]
        seg += [("text", tail, True), ("tok", EOS, True)]
        samples.append(pack_sample(seg, base_vocab, P, rng))
    return collate(samples, image_size, 5000 + 1000 * rank)


def interleave_batch(B=1, S=8192, n_images=4, base_vocab=32000, P=576, image_size=336, rank=0):
    """BASELINE cfg 5 (MMC4-style): images spread through one long supervised document."""
    samples = []
    for b in range(B):
        rng = np.random.RandomState(300 + b + 1000 * rank)
        text_total = S - 2 - n_images * (P + 2)
        per = text_total // (n_images + 1)
        seg = [("tok", BOS, False)]
        used = 0
        for _ in range(n_images):
            seg += [("text", per, True), ("img",)]
            used += per
        seg += [("text", text_total - used, True), ("tok", EOS, True)]
        samples.append(pack_sample(seg, base_vocab, P, rng))
    return collate(samples, image_size, 7000 + 1000 * rank)
