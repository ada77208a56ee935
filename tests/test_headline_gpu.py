"""The HEADLINE size as a test (VERDICT r5 "missing #2"): the full model (24-layer ViT-L/14-336 + mlp projector + 32-layer Llama-7B, bf16, random
init from the build's generator) on ONE BASELINE cfg-3 batch - B = 8 interpair sequences x S = 4096, 6 x 336-px frames each - exactly what
`bench.py` times and nothing else ran until now.  Reference path: llama_mmgpt.py:53-112 on collator.py:12-34 batches.

No CPU oracle finishes eight S = 4096 sequences at full depth in test time (~100 s per sequence forward-only), so the step is held to what the
path itself guarantees at any size:
  * the batch loss = the mean of the eight B = 1 losses (every sequence scores the same 603 positions, llama_mmgpt.py:92-100), to 1e-3;
  * the scored-rows backward (engine.sparse_head: head + last decoder layer contract over the ~15 % of rows the loss scores) = the dense backward,
    cosine >= 0.99999 on five named gradients from the top to the bottom of both towers;
  * engine.mem_level 2 (normed operands re-derived, SwiGLU outputs of 16 layers recomputed) = level 0, cosine >= 0.9999;
  * the same step twice = the same bits, over the WHOLE 14 GB gradient arena;
  * the attention backward took its five-product form (dS spill) and the peak stays inside the 288 GB.
The shallow real-width model vs the oracle at this sequence length is tests/test_geometry_gpu.py; full depth vs the oracle at cfg 2 is there too.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

VT = "model.vision_tower.vision_tower.vision_model."
NAMES = ["lm_head.weight", "model.layers.31.mlp.down_proj.weight", "model.layers.0.self_attn.q_proj.weight", "model.projector.projector.weight",
         VT + "encoder.layers.0.mlp.fc1.weight"]


def _arena_digest(flat):
    """Order-independent 64-bit digest of every 16-bit word of an arena (chunked: 7 B elements)."""
    w = flat.view(torch.int16)
    tot = torch.zeros((), dtype=torch.int64, device=flat.device)
    tot2 = torch.zeros((), dtype=torch.int64, device=flat.device)
    step = 1 << 27
    for i in range(0, w.numel(), step):
        c = w[i: i + step].to(torch.int64)
        tot += c.sum()
        tot2 += (c * c).sum() + (c[::7] * 31).sum()
    return int(tot), int(tot2)


def test_full_model_one_cfg3_training_step_B8_S4096():
    import bench
    from merlin_amd import ops as O
    from merlin_amd import synth
    from merlin_amd.model.llama_mmgpt import build_synthetic_model

    import gc

    dev = torch.device("cuda", 0)
    gc.collect()
    O.release_attn_scratch()
    torch.cuda.empty_cache()  # (earlier tests of this process leave cached blocks behind: the step needs 255 of the 288 GB in large pieces)
    free_b, total_b = torch.cuda.mem_get_info(dev)
    assert total_b > 250e9, "needs the MI355X's 288 GB: activations of the full cfg-3 step stay resident (255 GB peak)"
    model = build_synthetic_model(bench.LLAMA_7B, bench.VIT_L_336, projector="mlp", dtype=torch.bfloat16, device=dev, seed=0)
    eng = model.engine
    eng.save_activations = True  # bench.py's default configuration (bench.py: `eng.save_activations = not args.recompute`): activations stay resident
    assert eng.sparse_head and eng.mem_level == 0 and eng.fp32_residual
    batch = synth.interpair_batch(B=8, S=4096)
    assert batch["input_ids"].shape == (8, 4096) and all(im.shape == (6, 3, 336, 336) for im in batch["images"])
    n_scored = (batch["labels"][:, 1:] != -100).sum(1)
    assert int(n_scored.min()) == int(n_scored.max()) == 603  # equal counts: mean of per-sequence means == batch mean
    db = dict(input_ids=batch["input_ids"].to(dev), attention_mask=batch["attention_mask"].to(dev), labels=batch["labels"].to(dev),
              images=[im.to(dev) for im in batch["images"]])
    params = dict(model.named_parameters())

    def train_pass():
        for p in params.values():
            p.grad = None
        out = model(**db)
        out.loss.backward()
        torch.cuda.synchronize()
        return float(out.loss), {k: params[k].grad.detach().clone() for k in NAMES}

    def cos(a, b):
        a, b = a.double().reshape(-1), b.double().reshape(-1)
        return float(a @ b / (a.norm() * b.norm()).clamp_min(1e-300))

    torch.cuda.reset_peak_memory_stats(dev)
    loss8, g0 = train_pass()
    assert O.LAST_ATTN_BWD_FORM == "five-product", O.LAST_ATTN_BWD_FORM
    peak_gb = torch.cuda.max_memory_allocated(dev) / 1e9
    d0 = _arena_digest(eng.arena.gflat)
    assert all(bool(torch.isfinite(g.float()).all()) and float(g.float().norm()) > 0 for g in g0.values())
    # ---- the same step again: same bits in all 7 B gradient words ----
    loss8b, g0b = train_pass()
    d0b = _arena_digest(eng.arena.gflat)
    assert loss8b == loss8 and d0b == d0 and all(torch.equal(g0[k], g0b[k]) for k in NAMES), (loss8, loss8b, d0, d0b)
    del g0b
    # ---- batch loss vs the eight B = 1 losses ----
    singles = []
    with torch.no_grad():
        for b in range(8):
            o1 = model(input_ids=db["input_ids"][b: b + 1], attention_mask=db["attention_mask"][b: b + 1], labels=db["labels"][b: b + 1],
                       images=[db["images"][b]])
            singles.append(float(o1.loss))
    mean1 = sum(singles) / 8
    print(f"[cfg 3 full size] loss B=8 {loss8:.6f}; mean of eight B=1 losses {mean1:.6f} (min {min(singles):.4f} max {max(singles):.4f}); peak HBM {peak_gb:.1f} GB")
    assert abs(loss8 - mean1) < 1e-3 * abs(mean1), (loss8, mean1)
    # ---- scored-rows backward vs the dense backward ----
    eng.sparse_head = False
    try:
        loss_d, gd = train_pass()
    finally:
        eng.sparse_head = True
    rep = {k: (cos(g0[k], gd[k]), float(g0[k].float().norm() / gd[k].float().norm())) for k in NAMES}
    print("[cfg 3 full size] scored-rows vs dense backward (cosine, norm ratio):", rep)
    assert loss_d == loss8
    assert all(c >= 0.99999 and abs(r - 1) < 1e-3 for c, r in rep.values()), rep
    del gd
    # ---- mem_level 2 vs 0 ----
    eng.mem_level = 2
    try:
        loss_m, gm = train_pass()
    finally:
        eng.mem_level = 0
    rep = {k: (cos(g0[k], gm[k]), float(g0[k].float().norm() / gm[k].float().norm())) for k in NAMES}
    print("[cfg 3 full size] mem_level 2 vs 0 (cosine, norm ratio):", rep)
    assert abs(loss_m - loss8) <= 1e-6 * abs(loss8)
    assert all(c >= 0.9999 and abs(r - 1) < 5e-3 for c, r in rep.values()), rep
    assert peak_gb < 280.0, peak_gb
    del model, eng, params, g0, gm, db
    gc.collect()
    O.release_attn_scratch()
    torch.cuda.empty_cache()


def test_full_model_one_cfg5_fp8_training_step_B4_S8192():
    """BASELINE configs[4] at FULL size on one GPU (the 8-GPU leg is plain DP over it): B = 4 interleave documents x S = 8192 (4 images each, MMC4-style), full-depth
    ViT-L/14-336 + Llama-7B, every decoder / CLIP-tower Linear and the lm_head on the scaled-fp8 MFMA (forward, dgrad, wgrad) exactly as `bench.py`'s `extras.cfg5` runs
    it.  No reference counterpart exists for the fp8 path (SURVEY 8d cfg 5) and no CPU oracle finishes this size in test time (the S = 8192 sequence vs the oracle is
    tests/test_fp8_training_gpu.py on the real-width 2 + 2-layer model), so the step is held to size-independent properties:
      * finite loss near ln(vocab) for random-init weights, finite non-zero gradients at both ends of both towers;
      * the same step twice = the same bits over the whole 14-GB gradient arena (the fp8 step is deterministic: order-independent atomicMax maxima, fixed-order sums);
      * the fp8 batch loss agrees with the bf16 step's loss on the same batch to 2 % (e4m3 operands, per-row scales), and with the mean of the four B = 1 fp8 losses to
        1e-3 when every document scores the same number of positions (else to the token-weighted mean);
      * peak HBM inside the 288 GB."""
    import gc
    import math

    import bench
    from merlin_amd import ops as O
    from merlin_amd import synth
    from merlin_amd.model.llama_mmgpt import build_synthetic_model

    dev = torch.device("cuda", 0)
    gc.collect()
    O.release_attn_scratch()
    torch.cuda.empty_cache()
    model = build_synthetic_model(bench.LLAMA_7B, bench.VIT_L_336, projector="mlp", dtype=torch.bfloat16, device=dev, seed=0)
    eng = model.engine
    eng.save_activations = True
    batch = synth.interleave_batch(B=4, S=8192, n_images=4, rank=0)
    assert batch["input_ids"].shape == (4, 8192) and all(im.shape == (4, 3, 336, 336) for im in batch["images"])
    db = dict(input_ids=batch["input_ids"].to(dev), attention_mask=batch["attention_mask"].to(dev), labels=batch["labels"].to(dev),
              images=[im.to(dev) for im in batch["images"]])
    n_scored = (batch["labels"][:, 1:] != -100).sum(1)
    params = dict(model.named_parameters())

    def train_pass():
        for p in params.values():
            p.grad = None
        out = model(**db)
        out.loss.backward()
        torch.cuda.synchronize()
        return float(out.loss.detach())

    # ---- bf16 step on the same batch (16-bit residual streams, as the fp8 path keeps them) ----
    eng.fp32_residual = False
    loss16 = train_pass()
    # ---- the fp8 step, twice ----
    model.fp8_training = True
    tower_was, eng.fp8_tower = eng.fp8_tower, True
    try:
        torch.cuda.reset_peak_memory_stats(dev)
        loss8 = train_pass()
        peak_gb = torch.cuda.max_memory_allocated(dev) / 1e9
        assert dict(getattr(eng, "last_fp8", {})) == {"decoder": True, "tower": True, "head": True}, getattr(eng, "last_fp8", None)
        d0 = _arena_digest(eng.arena.gflat)
        g0 = {k: params[k].grad.detach().clone() for k in NAMES}
        loss8b = train_pass()
        d0b = _arena_digest(eng.arena.gflat)
        assert loss8b == loss8 and d0b == d0 and all(torch.equal(g0[k], params[k].grad) for k in NAMES), (loss8, loss8b, d0, d0b)
        assert all(bool(torch.isfinite(g.float()).all()) and float(g.float().norm()) > 0 for g in g0.values())
        singles = []
        with torch.no_grad():
            for b in range(4):
                o1 = model(input_ids=db["input_ids"][b: b + 1], attention_mask=db["attention_mask"][b: b + 1], labels=db["labels"][b: b + 1],
                           images=[db["images"][b]])
                singles.append(float(o1.loss))
    finally:
        model.fp8_training = False
        eng.fp8_tower = tower_was
        eng.fp32_residual = True
    w = n_scored.double() / n_scored.sum()
    mean1 = float(sum(wi * s for wi, s in zip(w.tolist(), singles)))
    print(f"[cfg 5 full size, fp8] loss B=4 {loss8:.6f}; token-weighted mean of four B=1 losses {mean1:.6f}; bf16 step on the same batch {loss16:.6f}; "
          f"scored positions per document {n_scored.tolist()}; peak HBM {peak_gb:.1f} GB")
    assert math.isfinite(loss8) and abs(loss8 - math.log(32003)) < 1.5, loss8
    assert abs(loss8 - mean1) < 1e-3 * abs(mean1), (loss8, mean1)
    assert abs(loss8 - loss16) < 2e-2 * abs(loss16), (loss8, loss16)
    assert peak_gb < 280.0, peak_gb
    del model, eng, params, g0, db
    gc.collect()
    O.release_attn_scratch()
    torch.cuda.empty_cache()
