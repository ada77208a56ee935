"""Pin the CPU oracle (oracle/ref_cpu.py) to outputs of the REAL reference captured by
oracle/make_golden.py (tests/golden/*.npz): logits, loss and every parameter gradient."""
import os

import numpy as np
import pytest
import torch

from oracle import cases as C
from oracle import ref_cpu as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    path = os.path.join(GOLD, name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated")
    return np.load(path)


@pytest.mark.parametrize("name", C.TINY_CASES)
def test_oracle_matches_reference_tiny(name):
    g = _load(name)
    cfg, batch = C.get_case(name)
    assert np.array_equal(g["input_ids"], batch["input_ids"].numpy())
    assert np.array_equal(g["labels"], batch["labels"].numpy())
    P = R.make_params(cfg, seed=0, requires_grad=True)
    loss, logits = R.forward(P, cfg, batch["input_ids"], batch["attention_mask"], batch["labels"], batch["images"])
    m = batch["attention_mask"].numpy()
    ref = g["logits"]
    got = logits.detach().numpy()
    # compare valid (non-pad) rows only
    err = np.abs(got - ref)[m].max() / np.abs(ref[m]).max()
    assert err < 2e-6, err
    assert abs(float(loss) - float(g["loss"])) < 2e-6 * abs(float(g["loss"]))
    loss.backward()
    checked = 0
    for n, p in P.items():
        key = f"grad/{n}/norm"
        if key not in g.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        d = R_digest(p.grad.numpy())
        assert abs(d["norm"] - float(g[key])) <= 1e-5 * float(g[key]) + 1e-9, (n, d["norm"], float(g[key]))
        scale = max(1e-9, np.abs(g[f"grad/{n}/strided"]).max())
        assert np.abs(d["strided"] - g[f"grad/{n}/strided"]).max() <= 2e-5 * scale + 1e-9, n
        checked += 1
    assert checked > 40


def R_digest(a):
    f = a.reshape(-1).astype(np.float64)
    stride = max(1, f.size // 257)
    return {"norm": float(np.sqrt((f * f).sum())), "strided": f[::stride][:512].astype(np.float32)}


def test_oracle_matches_reference_medium():
    g = _load("medium_cfg1")
    cfg, batch = C.get_case("medium_cfg1")
    P = R.make_params(cfg, seed=0)
    with torch.no_grad():
        loss, logits = R.forward(P, cfg, batch["input_ids"], batch["attention_mask"], batch["labels"], batch["images"])
    got = logits.numpy()[:, ::8, :512]
    ref = g["logits_slice"]
    assert np.abs(got - ref).max() / float(g["logits_absmax"]) < 5e-6
    lse = torch.logsumexp(logits, dim=-1).numpy()
    assert np.abs(lse - g["logits_lse"]).max() < 1e-4
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))


def test_oracle_matches_reference_released_geometry():
    """448 px / conv stride 2 / P = 256 at real widths (the released checkpoint's geometry, pretrain.sh:6-9)."""
    g = _load("released_conv448")
    cfg, batch = C.get_case("released_conv448")
    assert cfg.num_patches == 256 and np.array_equal(g["input_ids"], batch["input_ids"].numpy())
    P = R.make_params(cfg, seed=0)
    with torch.no_grad():
        loss, logits = R.forward(P, cfg, batch["input_ids"], batch["attention_mask"], batch["labels"], batch["images"])
    got = logits.numpy()[:, ::4, :512]
    assert np.abs(got - g["logits_slice"]).max() / float(g["logits_absmax"]) < 5e-6
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))


def test_splice_errors_match_reference_messages():
    """base_mmgpt.py:116-118,125-126: ValueError on start/end mismatch or misplaced <im_end>."""
    cfg, batch = C.get_case("tiny_1img")
    P = R.make_params(cfg, seed=0)
    ids = batch["input_ids"].clone()
    end = int((ids[0] == cfg.im_end_token).nonzero()[0])
    bad = ids.clone()
    bad[0, end] = 5
    with pytest.raises(ValueError):
        R.forward(P, cfg, bad, batch["attention_mask"], batch["labels"], batch["images"])
    bad = ids.clone()
    bad[0, end], bad[0, end + 1] = bad[0, end + 1].item(), cfg.im_end_token
    with pytest.raises(ValueError):
        R.forward(P, cfg, bad, batch["attention_mask"], batch["labels"], batch["images"])
