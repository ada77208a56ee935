"""End-to-end parity of the HIP path (through the reference's model surface) against
  (a) golden vectors captured from the REAL reference (tests/golden/*.npz, oracle/make_golden.py) and
  (b) the CPU oracle (oracle/ref_cpu.py) run on the same generated weights for gradients.

Tolerances (stated, per dtype): logits  max|d| / max|ref|  <= 1e-3 (fp16: BASELINE.json's "within 1e-3 rel fp16", asserted on the
tiny, medium and released-geometry models in the DEFAULT configuration = fp32 residual streams) / 1.5e-2 (bf16); loss rel-err <= 1e-3 / 5e-3; every parameter-gradient tensor: cosine >= 0.999
(fp16) / 0.995 (bf16) and norm ratio within 1% / 3%.  BASELINE.json's "1e-3 rel fp16" is the fp16 row.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
# fp16 = BASELINE.json's "within 1e-3 rel fp16" row: asserted at 1e-3 on the tiny fixtures, the medium model (real widths; measured 8.5e-4,
# profiles/r04_parity_floor.txt) and the released geometry in the default configuration (fp32 residual streams).  With 16-bit streams
# (engine.fp32_residual = False, the reference's own bf16-style storage) real widths cost more (medium 1.5e-3, released 1.2e-3): `logits_16bit_stream`.
# Full 7B depth sits on the measured 16-bit-operand floor (tests/test_parity_floor_gpu.py) and at 1e-4 in the fp32-store parity mode.
TOL = {torch.float16: dict(logits=1e-3, logits_tiny=1e-3, logits_16bit_stream=2e-3, loss=1e-3, cos=0.999, norm=0.01),
       torch.bfloat16: dict(logits=1.5e-2, loss=5e-3, cos=0.995, norm=0.03)}


def _build(cfg, dtype, **kw):
    from merlin_amd.model.llama_mmgpt import build_synthetic_model

    llama = dict(vocab_size=cfg.vocab_size - 3, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                 num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                 rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, max_position_embeddings=8192)
    vision = dict(hidden_size=cfg.v_hidden_size, intermediate_size=cfg.v_intermediate_size, num_hidden_layers=cfg.v_num_hidden_layers,
                  num_attention_heads=cfg.v_num_attention_heads, image_size=cfg.v_image_size, patch_size=cfg.v_patch_size,
                  layer_norm_eps=cfg.v_layer_norm_eps)
    m = build_synthetic_model(llama, vision, projector=cfg.projector, conv_stride=cfg.conv_stride, dtype=dtype, device="cuda", seed=0, **kw)
    assert (m.im_patch_token, m.im_start_token, m.im_end_token) == (cfg.im_patch_token, cfg.im_start_token, cfg.im_end_token)
    return m


def _to_dev(batch):
    return dict(input_ids=batch["input_ids"].cuda(), attention_mask=batch["attention_mask"].cuda(), labels=batch["labels"].cuda(),
                images=[im.cuda() for im in batch["images"]])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name", ["tiny_1img", "tiny_2img", "tiny_padbatch", "tiny_textonly", "tiny_conv2"])
def test_tiny_forward_backward_parity(name, dtype):
    from oracle import cases as C
    from oracle import ref_cpu as R

    tol = TOL[dtype]
    cfg, batch = C.get_case(name)
    g = np.load(os.path.join(GOLD, name + ".npz"))
    model = _build(cfg, dtype)
    # weights on the device are the generator's bits
    P = R.make_params(cfg, seed=0, requires_grad=True)
    sd = dict(model.named_parameters())
    assert set(sd) == set(P), set(sd) ^ set(P)
    for k in ("model.layers.1.mlp.up_proj.weight", "model.norm.weight", "lm_head.weight",
              "model.vision_tower.vision_tower.vision_model.encoder.layers.0.self_attn.q_proj.bias"):
        assert torch.equal(sd[k].detach().float().cpu(), P[k].detach()), k
    out = model(**_to_dev(batch))
    logits = out.logits.float().cpu().numpy()
    mask = batch["attention_mask"].numpy()
    ref = g["logits"]
    err = np.abs(logits - ref)[mask].max() / np.abs(ref[mask]).max()
    assert err < tol.get("logits_tiny", tol["logits"]), f"logits rel err {err}"
    assert abs(float(out.loss) - float(g["loss"])) < tol["loss"] * abs(float(g["loss"]))
    # padded query rows: the flash path returns finite values (zeros from attention)
    assert np.isfinite(logits).all()
    out.loss.backward()
    loss_ref, _ = R.forward(P, cfg, batch["input_ids"], batch["attention_mask"], batch["labels"], batch["images"])
    loss_ref.backward()
    bad = []
    for k, p in model.named_parameters():
        gr = P[k].grad
        if gr is None or float(gr.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.float().abs().max()) == 0.0, f"{k}: expected zero grad"
            continue
        assert p.grad is not None, k
        if k.endswith("self_attn.k_proj.bias"):
            # d(loss)/d(k bias) is exactly 0 in exact arithmetic (softmax is invariant to adding q.b to every
            # key's score): both sides are rounding noise.  Require ours to be noise-sized vs the q bias grad.
            qb = sd[k.replace("k_proj", "q_proj")].grad.float().norm()
            assert float(p.grad.float().norm()) < 2e-2 * float(qb), k
            continue
        a, b = p.grad.float().cpu().reshape(-1).double(), gr.reshape(-1).double()
        cos = float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-30))
        ratio = float(a.norm() / b.norm())
        if cos < tol["cos"] or abs(ratio - 1) > tol["norm"]:
            bad.append((k, cos, ratio))
    assert not bad, bad[:8]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_medium_forward_parity_vs_reference_golden(dtype):
    """Real head dims / widths (2+2 layers, S=613): logits slice, per-position logsumexp and loss vs the
    reference's outputs captured in tests/golden/medium_cfg1.npz."""
    from oracle import cases as C

    tol = TOL[dtype]
    cfg, batch = C.get_case("medium_cfg1")
    g = np.load(os.path.join(GOLD, "medium_cfg1.npz"))
    model = _build(cfg, dtype)
    with torch.no_grad():
        out = model(**_to_dev(batch))
    lg = out.logits.float()
    got = lg[:, ::8, :512].cpu().numpy()
    err = np.abs(got - g["logits_slice"]).max() / float(g["logits_absmax"])
    assert err < tol["logits"], err
    lse = torch.logsumexp(lg, dim=-1).cpu().numpy()
    assert np.abs(lse - g["logits_lse"]).max() < 10 * tol["logits"]
    assert abs(float(out.loss) - float(g["loss"])) < tol["loss"] * abs(float(g["loss"]))


@pytest.mark.parametrize("stream", ["16bit", "fp32"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_full_7b_cfg1_forward_parity_vs_reference_golden(dtype, stream):
    """BASELINE cfg 1/2 at FULL size (24-layer ViT-L/14-336 + 32-layer Llama-7B, S=613): logits slice,
    per-position logsumexp and loss of the HIP path vs the REAL reference's fp32 CPU outputs
    (tests/golden/full_cfg1.npz, generated by oracle/make_golden.py full).  Tolerance at full depth (32 layers
    of 16-bit residual stream): logits max 5e-3 / rms 1.25e-3 (fp16), 5e-2 / 1e-2 (bf16) of max|logit|, loss 2e-3 / 1e-2; with the fp32
    residual stream (the engine's default) 2.6e-3 / 6e-4 and 2.2e-2 / 4.8e-3."""
    from oracle import cases as C

    path = os.path.join(GOLD, "full_cfg1.npz")
    assert os.path.exists(path), "tests/golden/full_cfg1.npz is a committed fixture (oracle/make_golden.py full)"
    g = np.load(path)
    cfg, batch = C.get_case("full_cfg1")
    assert np.array_equal(g["input_ids"], batch["input_ids"].numpy())
    model = _build(cfg, dtype)
    # stream = "fp32": engine.fp32_residual - both towers' residual streams in fp32 with the PRODUCTION GEMM / attention / norm kernels
    # (VERDICT r2 #4); what is left is the 16-bit rounding of the GEMM operands (measured: profiles/r03_skinny_gemm.txt, profiles/r04_parity_floor.txt)
    model.engine.fp32_residual = stream == "fp32"
    with torch.no_grad():
        out = model(**_to_dev(batch))
    lg = out.logits.float()
    got = lg[:, ::16, :256].cpu().numpy()
    # Two statistics.  The rms error over the slice is stable to the last digit under any change of summation order (measured, one-pass
    # GEMMs / split-K for the skinny projections: fp16 1.066e-3 / 1.072e-3, fp16 + fp32 stream 5.210e-4 / 5.203e-4, bf16 8.50e-3 / 8.49e-3,
    # bf16 + fp32 stream 4.16e-3 / 4.10e-3) and is held within 15 %.  The maximum is ONE element of 40 k and scatters by +-25 % with
    # rounding order (same pairs: 4.59e-3 / 5.08e-3, 2.11e-3 / 2.65e-3, 3.49e-2 / 3.26e-2, 1.76e-2 / 1.93e-2).  Its bounds are the ones the
    # one-pass summation order was first measured under (5e-3 / 2.6e-3 fp16, 5e-2 / 2.2e-2 bf16): fp16 - the dtype BASELINE's tolerance is
    # stated in - keeps that order (ops._skinny_splitk_ok: split-K of the skinny projections is a bf16-only launch plan), bf16 meets
    # them with split-K.  tests/test_parity_floor_gpu.py holds the same errors against the measured 16-bit-operand floor.
    tl, trms, tloss = (5e-3, 1.25e-3, 2e-3) if dtype == torch.float16 else (5e-2, 1.0e-2, 1e-2)
    if stream == "fp32":
        # (round 4: the streams start from fp32 tensors and RoPE / SwiGLU run on the fp32 accumulators - measured 2.10e-3 / 4.03e-4 and 1.46e-2 / 3.22e-3,
        #  i.e. the 16-bit-operand floor itself; the rms bounds follow with 15 % margin, the single-element maxima keep their round-3 bounds)
        tl, trms = (2.6e-3, 4.7e-4) if dtype == torch.float16 else (2.2e-2, 3.8e-3)
    dlt = got - g["logits_slice"]
    err = np.abs(dlt).max() / float(g["logits_absmax"])
    rms = float(np.sqrt((dlt.astype(np.float64) ** 2).mean())) / float(g["logits_absmax"])
    assert err < tl and rms < trms, (err, rms)
    lse = torch.logsumexp(lg, dim=-1).cpu().numpy()
    assert np.abs(lse - g["logits_lse"]).max() < 10 * tl
    assert abs(float(out.loss) - float(g["loss"])) < tloss * abs(float(g["loss"])), (float(out.loss), float(g["loss"]))
    print(f"[full cfg1 {dtype} residual stream {stream}] logits rel err max {err:.3e} rms {rms:.3e}  loss hip {float(out.loss):.5f} ref {float(g['loss']):.5f}")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name", ["tiny_2img", "tiny_padbatch", "tiny_conv2"])
def test_fp32_residual_stream_forward_backward_parity(name, dtype):
    """engine.fp32_residual on the tiny fixtures: forward + backward vs the fp32 oracle (the backward runs on the 16-bit copies of the
    layer inputs the stream's reader emits), with activations resident and with layer recompute; never worse than the 16-bit stream."""
    from oracle import cases as C
    from oracle import ref_cpu as R

    cfg, batch = C.get_case(name)
    P = R.make_params(cfg, seed=0, requires_grad=True)
    loss_ref, logits_ref = R.forward(P, cfg, batch["input_ids"], batch["attention_mask"], batch["labels"], batch["images"])
    loss_ref.backward()
    valid = batch["attention_mask"].bool()
    want = logits_ref.detach()[valid]
    model = _build(cfg, dtype)
    errs = {}
    for r32, save in ((False, True), (True, True), (True, False)):
        for p in model.parameters():
            p.grad = None
        model.engine.fp32_residual, model.engine.save_activations = r32, save
        out = model(**_to_dev(batch))
        out.loss.backward()
        errs[(r32, save)] = float((out.logits.float().cpu()[valid] - want).abs().max() / want.abs().max())
        if r32:
            named = dict(model.named_parameters())
            bad = []
            for k in ("model.layers.0.self_attn.q_proj.weight", "model.layers.1.mlp.down_proj.weight", "model.layers.0.input_layernorm.weight",
                      "lm_head.weight", "model.projector.projector.weight",
                      "model.vision_tower.vision_tower.vision_model.encoder.layers.0.mlp.fc2.weight",
                      "model.vision_tower.vision_tower.vision_model.encoder.layers.0.layer_norm1.weight"):
                if named[k].grad is None:
                    continue
                gg, gr = named[k].grad.float().cpu().reshape(-1), P[k].grad.reshape(-1)
                cos = float(gg @ gr / (gg.norm() * gr.norm() + 1e-30))
                if cos < (0.999 if dtype == torch.float16 else 0.99):
                    bad.append((k, cos))
            assert not bad, bad
            assert abs(float(out.loss) - float(loss_ref)) < (2e-3 if dtype == torch.float16 else 2e-2) * abs(float(loss_ref))
    print(f"[fp32 residual {name} {dtype}] logits rel err 16-bit stream {errs[(False, True)]:.3e} -> fp32 stream {errs[(True, True)]:.3e}")
    # (resident vs recompute differ in the tower only: fc1 + quick-GELU is one fused launch when nothing is kept)
    assert errs[(True, True)] <= 1.05 * errs[(False, True)] + 1e-5 and abs(errs[(True, True)] - errs[(True, False)]) < (2e-4 if dtype == torch.float16 else 2e-3)
    assert errs[(True, True)] < (1e-3 if dtype == torch.float16 else 1e-2)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_lm_head_backward_over_the_scored_rows_only(dtype):
    """engine.sparse_head (+ sparse_last_layer): positions whose shifted label is -100 have exactly zero rows in dlogits, so the head's dgrad /
    wgrad contract over the scored rows alone (compacted on the device) when those are at most half of the batch - and so do the MLP half, the
    post-attention norm and the o projection of the LAST decoder layer, whose output gradient is zero on the same rows.  Interpair-style batch (labels on the trajectory tail
    only, like cfg 3): every gradient against the dense products (cosine >= 0.99999, max |diff| <= 2 % of max |g| - fp32 sums of the same non-zero
    terms in another order, one 16-bit rounding each) and against the fp32 oracle with the usual bounds; the loss is the same number."""
    from merlin_amd import synth
    from oracle import cases as C
    from oracle import ref_cpu as R

    tol = TOL[dtype]
    cfg = C.tiny_cfg()
    batch = synth.interpair_batch(B=4, S=512, frames=12, base_vocab=cfg.vocab_size - 3, P=cfg.num_patches, image_size=cfg.v_image_size)
    n_scored = int((batch["labels"][:, 1:] != -100).sum())
    assert batch["input_ids"].shape == (4, 512) and 0 < 2 * ((n_scored + 255) // 256 * 256) <= 2048, n_scored  # (988 scored rows of 2048)
    model = _build(cfg, dtype)
    grads, losses = {}, {}
    for sparse in (False, True):
        for p in model.parameters():
            p.grad = None
        model.engine.sparse_head = model.engine.sparse_last_layer = sparse
        out = model(**_to_dev(batch))
        out.loss.backward()
        grads[sparse] = {k: p.grad.detach().float().cpu() for k, p in model.named_parameters() if p.grad is not None}
        losses[sparse] = float(out.loss)
    assert losses[True] == losses[False]
    bad = []
    for k, g in grads[False].items():
        a = grads[True][k]
        if float(g.abs().max()) == 0.0:
            assert float(a.abs().max()) == 0.0, k
            continue
        cos = float((a.reshape(-1).double() @ g.reshape(-1).double()) / (a.double().norm() * g.double().norm()))
        md = float((a - g).abs().max() / g.abs().max())
        if cos < 0.99999 or md > 0.02:
            bad.append((k, cos, md))
    assert not bad, bad[:6]
    P = R.make_params(cfg, seed=0, requires_grad=True)
    loss_ref, _ = R.forward(P, cfg, batch["input_ids"], batch["attention_mask"], batch["labels"], batch["images"])
    loss_ref.backward()
    bad = []
    for k, a in grads[True].items():  # against the oracle: never further away than the dense products (the CLIP q / k gradients of 48 tiny frames are noisy in both)
        gr = P[k].grad
        if gr is None or float(gr.abs().max()) == 0.0 or k.endswith("self_attn.k_proj.bias"):
            continue
        cs = lambda x: float((x.reshape(-1).double() @ gr.reshape(-1).double()) / (x.double().norm() * gr.double().norm()).clamp_min(1e-30))  # noqa: E731
        c_sparse, c_dense = cs(a), cs(grads[False][k])
        if c_sparse < min(tol["cos"], c_dense - 1e-3):
            bad.append((k, c_sparse, c_dense))
    assert not bad, bad[:6]
    assert abs(losses[True] - float(loss_ref)) < tol["loss"] * abs(float(loss_ref))
    # the compact last layer re-derives its operands from compact rows under mem_level 2 as well
    for p in model.parameters():
        p.grad = None
    model.engine.mem_level = 2
    try:
        model(**_to_dev(batch)).loss.backward()
    finally:
        model.engine.mem_level = 0
    for k, g in grads[True].items():
        a = model.get_parameter(k).grad.detach().float().cpu()
        if float(g.abs().max()) == 0.0:
            continue
        cos = float((a.reshape(-1).double() @ g.reshape(-1).double()) / (a.double().norm() * g.double().norm()))
        assert cos >= 0.9999, (k, cos)


def test_scored_rows_count_survives_many_outstanding_forwards():
    """the scored-row count of a grad-enabled forward is read back in ITS backward.  Twenty forwards with DIFFERENT scored-row counts
    are left outstanding (their losses summed, one .backward()): every backward must use its own count - the gradients equal those of the dense
    head on the same twenty micro-batches.  (A shared ring of 16 slots silently dropped scored rows of forwards 0-3 here.)"""
    from merlin_amd import synth
    from oracle import cases as C

    cfg = C.tiny_cfg()
    model = _build(cfg, torch.bfloat16)
    batches = []
    for i in range(20):
        b = synth.interpair_batch(B=4, S=512, frames=12, base_vocab=cfg.vocab_size - 3, P=cfg.num_patches, image_size=cfg.v_image_size, rank=i)
        lab = b["labels"].clone()
        lab[:, : 512 - 20 - 11 * i] = -100  # 4 x (19 + 11 i) scored positions: another count (and another 256-row bucket) per forward
        b["labels"] = lab
        batches.append(_to_dev(b))
    counts = [int((b["labels"][:, 1:] != -100).sum()) for b in batches]
    assert len(set(counts)) == 20 and len({(c + 255) // 256 for c in counts}) >= 4 and 2 * ((max(counts) + 255) // 256 * 256) <= 2048, counts
    grads = {}
    for sparse in (False, True):
        for p in model.parameters():
            p.grad = None
        model.engine.sparse_head = model.engine.sparse_last_layer = sparse
        total = None
        for b in batches:
            loss = model(**b).loss
            total = loss if total is None else total + loss
        total.backward()
        grads[sparse] = {k: p.grad.detach().float().cpu() for k, p in model.named_parameters() if p.grad is not None}
    model.engine.sparse_head = model.engine.sparse_last_layer = True
    bad = []
    for k, g in grads[False].items():
        a = grads[True][k]
        if float(g.abs().max()) == 0.0:
            continue
        cos = float((a.reshape(-1).double() @ g.reshape(-1).double()) / (a.double().norm() * g.double().norm()))
        if cos < 0.9999 or abs(float(a.double().norm() / g.double().norm()) - 1) > 5e-3:
            bad.append((k, cos, float(a.double().norm() / g.double().norm())))
    assert not bad, bad[:6]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name", ["tiny_2img", "tiny_padbatch"])
def test_mem_level_rederived_activations_give_the_same_gradients(name, dtype):
    """engine.mem_level 1 / 2 (bench.py's slope between 'everything resident' and full layer recompute): the normed GEMM operands h1 / h2 are
    re-derived in backward from the saved 16-bit layer inputs, act = silu(gate) * up from the saved gate|up tensor.  With 16-bit residual
    streams that is the forward's own arithmetic: gradients bit-identical to mem_level 0.  With fp32 streams (default) the forward normed the
    fp32 stream and the backward norms its 16-bit copy - one rounding of the norm's INPUT apart: every gradient within cosine 0.9999."""
    from oracle import cases as C

    cfg, batch = C.get_case(name)
    model = _build(cfg, dtype)
    eng = model.engine
    eng.save_activations = True
    eng.mem_act_layers = 1
    grads = {}
    for r32 in (False, True):
        for lvl in (0, 1, 2):
            for p in model.parameters():
                p.grad = None
            eng.fp32_residual, eng.mem_level = r32, lvl
            out = model(**_to_dev(batch))
            out.loss.backward()
            grads[(r32, lvl)] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    eng.mem_level = 0
    assert all(torch.equal(grads[(False, 1)][k], g) for k, g in grads[(False, 0)].items()), "16-bit streams, mem_level 1"
    # (mem_level 2: the 4-wave GEMM's fused SwiGLU gates the fp32 accumulators, the stand-alone kernel the rounded gate|up tensor - one rounding apart
    #  at shapes that kernel takes)
    for r32, lvl in ((False, 2), (True, 1), (True, 2)):
        bad = []
        for k, g in grads[(r32, 0)].items():
            a, b = grads[(r32, lvl)][k].float().reshape(-1), g.float().reshape(-1)
            if float(b.norm()) == 0.0:
                continue
            cos = float(a @ b / (a.norm() * b.norm()).clamp_min(1e-30))
            if cos < 0.9999:
                bad.append((k, cos))
        assert not bad, (r32, lvl, bad[:6])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_medium_backward_grad_norms_vs_reference_golden(dtype):
    """Gradient digests (norm + strided sample) of every parameter vs the reference's (medium golden).
    fp16: every sampled cosine >= 0.999 and norms within 1 % - the strict check of the backward maths.
    bf16 (8x coarser mantissa): cosine >= 0.99, norms within 5 %; the 512-element strided sample of lm_head's gradient is
    the exception: it consists of non-label vocabulary rows whose entries are sums of ~33 softmax probabilities
    p ~ 3e-5, and bf16 logit noise (|d logit| ~ 0.1) moves each p by ~10 %, so that 512-element sample's cosine is noise
    (measured anywhere from 0.01 to 0.96 depending on rounding details upstream; 0.9999 in fp16) - its norm is pinned tightly and the
    WHOLE tensor is held against the CPU ORACLE's lm_head gradient (fp32 autograd of oracle/ref_cpu.py on the same weights and batch:
    an oracle check in both dtypes, not a comparison of two HIP runs)."""
    from oracle import cases as C
    from oracle import ref_cpu as R

    cfg, batch = C.get_case("medium_cfg1")
    g = np.load(os.path.join(GOLD, "medium_cfg1.npz"))
    model = _build(cfg, dtype)
    out = model(**_to_dev(batch))
    out.loss.backward()
    strict = dtype == torch.float16
    bad = []
    n = 0
    for k, p in model.named_parameters():
        key = f"grad/{k}/norm"
        if key not in g.files or float(g[key]) == 0.0 or k.endswith("self_attn.k_proj.bias"):  # k bias: true grad is 0
            continue
        f = p.grad.float().reshape(-1)
        stride = max(1, f.numel() // 257)
        samp = f[::stride][:512].cpu().numpy().astype(np.float64)
        ref = g[f"grad/{k}/strided"].astype(np.float64)
        cos = float(samp @ ref / max(1e-30, np.linalg.norm(samp) * np.linalg.norm(ref)))
        ratio = float(f.double().norm()) / float(g[key])
        n += 1
        if strict:
            cmin, rtol = 0.999, 0.01
        else:
            cmin, rtol = (-1.0 if k == "lm_head.weight" else 0.99), 0.05  # (lm_head's bf16 SAMPLE is noise: whole tensor vs the oracle below)
        if cos < cmin or abs(ratio - 1) > rtol:
            bad.append((k, cos, ratio))
    assert n > 30
    assert not bad, bad[:8]
    # lm_head.weight's gradient, whole tensor, against the oracle's (only that leaf requires grad: autograd stops at the head)
    gh = dict(model.named_parameters())["lm_head.weight"].grad.float().cpu()
    P = R.make_params(cfg, seed=0)
    P["lm_head.weight"].requires_grad_(True)
    loss_ref, _ = R.forward(P, cfg, batch["input_ids"], batch["attention_mask"], batch["labels"], batch["images"])
    loss_ref.backward()
    gr = P["lm_head.weight"].grad
    assert abs(float(gr.double().norm()) / float(g["grad/lm_head.weight/norm"]) - 1) < 1e-4  # the oracle's gradient IS the reference's
    cos = float((gh.double() * gr.double()).sum() / (gh.double().norm() * gr.double().norm()))
    ratio = float(gh.double().norm() / gr.double().norm())
    print(f"[medium lm_head.weight grad {dtype}] whole-tensor cosine vs oracle {cos:.5f} norm ratio {ratio:.4f}")
    assert cos > (0.9995 if strict else 0.97) and abs(ratio - 1) < (0.01 if strict else 0.05), (cos, ratio)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_token_count_multiple_of_64_path(dtype):
    """B*S % 64 == 0 (like BASELINE cfg 3): the backward uses the K-strided (transpose-read) wgrad / dgrad GEMMs
    instead of re-laid-out copies.  Interpair-style packed batch (2 frames per sequence) vs the CPU oracle."""
    from merlin_amd import synth
    from oracle import cases as C
    from oracle import ref_cpu as R

    tol = TOL[dtype]
    cfg = C.tiny_cfg()
    batch = synth.interpair_batch(B=2, S=64, frames=2, base_vocab=cfg.vocab_size - 3, P=cfg.num_patches, image_size=cfg.v_image_size)
    assert batch["input_ids"].shape == (2, 64)
    model = _build(cfg, dtype)
    out = model(**_to_dev(batch))
    out.loss.backward()
    P = R.make_params(cfg, seed=0, requires_grad=True)
    loss_ref, logits_ref = R.forward(P, cfg, batch["input_ids"], batch["attention_mask"], batch["labels"], batch["images"])
    loss_ref.backward()
    err = float((out.logits.float().cpu() - logits_ref.detach()).abs().max() / logits_ref.detach().abs().max())
    assert err < tol["logits"], err
    bad = []
    for k, p in model.named_parameters():
        gr = P[k].grad
        if gr is None or float(gr.abs().max()) == 0.0 or k.endswith("self_attn.k_proj.bias"):
            continue
        a, b = p.grad.float().cpu().reshape(-1).double(), gr.reshape(-1).double()
        cos = float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-30))
        ratio = float(a.norm() / b.norm())
        if cos < tol["cos"] or abs(ratio - 1) > tol["norm"]:
            bad.append((k, cos, ratio))
    assert not bad, bad[:8]


def test_splice_errors_raise_like_reference():
    from oracle import cases as C

    cfg, batch = C.get_case("tiny_1img")
    model = _build(cfg, torch.bfloat16)
    b = _to_dev(batch)
    end = int((b["input_ids"][0] == cfg.im_end_token).nonzero()[0])
    bad = b["input_ids"].clone()
    bad[0, end] = 5
    with pytest.raises(ValueError):
        model(input_ids=bad, attention_mask=b["attention_mask"], labels=b["labels"], images=b["images"])
    bad = b["input_ids"].clone()
    bad[0, end], bad[0, end + 1] = bad[0, end + 1].item(), cfg.im_end_token
    with pytest.raises(ValueError):
        model(input_ids=bad, attention_mask=b["attention_mask"], labels=b["labels"], images=b["images"])


def test_grad_accumulation_and_zero_grad():
    """Two backward passes accumulate into the gradient arena; zero_grad(set_to_none) resets it."""
    from oracle import cases as C

    cfg, batch = C.get_case("tiny_1img")
    model = _build(cfg, torch.float16)
    b = _to_dev(batch)
    model(**b).loss.backward()
    g1 = {k: p.grad.float().clone() for k, p in model.named_parameters() if p.grad is not None}
    model(**b).loss.backward()
    for k, p in model.named_parameters():
        if k in g1 and float(g1[k].abs().max()) > 0:
            r = float((p.grad.float() - 2 * g1[k]).abs().max() / g1[k].abs().max())
            assert r < 5e-3, (k, r)
    for p in model.parameters():
        p.grad = None
    model(**b).loss.backward()
    for k, p in model.named_parameters():
        if k in g1 and float(g1[k].abs().max()) > 0:
            assert float((p.grad.float() - g1[k]).abs().max() / g1[k].abs().max()) < 1e-6, k


def test_frozen_tower_and_state_dict_keys():
    from oracle import cases as C
    from oracle import ref_cpu as R

    cfg, batch = C.get_case("tiny_1img")
    model = _build(cfg, torch.bfloat16, freeze_vision_tower=True)
    model.get_model().vision_tower.requires_grad_(False)
    assert set(model.state_dict().keys()) == set(R.param_shapes(cfg).keys())
    model(**_to_dev(batch)).loss.backward()
    for k, p in model.named_parameters():
        if "vision_tower" in k:
            assert p.grad is None
    assert model.lm_head.weight.grad is not None and float(model.lm_head.weight.grad.float().abs().max()) > 0


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name", ["tiny_1img", "tiny_padbatch"])
def test_kv_cache_decode_matches_full_recompute_and_oracle(name, dtype):
    """generate() with a KV cache (prefill + one HBM-bound decode step per token) against (a) the same model decoding by
    full-sequence recompute and (b) the fp32 CPU oracle run on the extended sequence, teacher-forced on the same tokens:
    the logits that pick every new token must agree (tolerance of the dtype), for a single prompt with an image and for
    a right-padded batch whose rows continue from their own lengths."""
    from oracle import cases as C
    from oracle import ref_cpu as R

    tol = TOL[dtype]["logits"] * 2
    cfg, batch = C.get_case(name)
    model = _build(cfg, dtype)
    dev_b = _to_dev(batch)
    ids, am, images = dev_b["input_ids"], dev_b["attention_mask"], dev_b["images"]
    n_new = 6
    logits, cache = model.engine.prefill(ids, am, images, n_new)
    lens = am.to(torch.bool).sum(dim=1)
    B = ids.shape[0]
    P = R.make_params(cfg, seed=0, requires_grad=False)
    cur_ids = [ids[b, :int(lens[b])].cpu() for b in range(B)]
    for step in range(n_new):
        nxt = logits.argmax(dim=-1)
        for b in range(B):
            # oracle: full forward of this row's sequence so far (no padding), last-position logits
            with torch.no_grad():
                ref = R.forward(P, cfg, cur_ids[b][None], torch.ones(1, len(cur_ids[b]), dtype=torch.bool), None,
                                [batch["images"][b]])[1][0, -1]
            got = logits[b].float().cpu()
            assert float((got - ref).abs().max() / ref.abs().max()) < tol, (name, step, b)
            cur_ids[b] = torch.cat([cur_ids[b], nxt[b:b + 1].cpu()])
        if step + 1 < n_new:
            logits = model.engine.decode_step(nxt, cache)
    assert torch.equal(cache.lens.cpu(), (lens + n_new - 1).to(torch.int32).cpu())
    # the public surface: cached and recompute decoding agree token for token (batch 1: no padding involved)
    if B == 1:
        a = model.generate(ids, images=images, max_new_tokens=n_new, use_cache=True, eos_token_id=-1)
        a_eager = model.generate(ids, images=images, max_new_tokens=n_new, use_cache=True, use_graph=False, eos_token_id=-1)
        assert torch.equal(a, a_eager), "graph replay and eager decode must be bit-identical"
        b_ = model.generate(ids, images=images, max_new_tokens=n_new, use_cache=False, eos_token_id=-1)
        assert a.shape == (1, ids.shape[1] + n_new)
        assert torch.equal(a[:, :ids.shape[1]], ids)
        # identical unless two candidates are within rounding of each other; require >= n_new-1 agreeing tokens
        assert int((a == b_).sum()) >= a.numel() - 1


def test_fused_adamw_llrd_clip_schedule_vs_torch_adamw():
    """One optimizer step of the fused arena AdamW with the reference's recipe (pretrain.sh:23-29: --llrd, lr 5e-5,
    beta2 0.95, wd 0.05, cosine warm-up; HF's default max_grad_norm clipping) against torch.optim.AdamW driven by the
    reference's param groups and torch.nn.utils.clip_grad_norm_ on fp32 copies of the same bf16 parameters/gradients:
    every parameter must land within one bf16 rounding of the fp32 result."""
    from oracle import cases as C
    from merlin_amd import optim as OPT

    cfg, batch = C.get_case("tiny_2img")
    dtype = torch.bfloat16
    model = _build(cfg, dtype)
    out = model(**_to_dev(batch))
    out.loss.backward()
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    lr, wd, clip_at = 5e-3, 0.05, 0.05
    mult = OPT.cosine_with_warmup(3, 100, 0.05)
    # fp32 reference
    master = {n: p.detach().float().clone().requires_grad_(True) for n, p in named}
    for n, p in named:
        master[n].grad = p.grad.detach().float().clone()
    from oracle import llrd_ref

    groups = llrd_ref.param_groups(named, lr * mult, wd, OPT.vit_lr_scale)
    ref_opt = torch.optim.AdamW([{"params": [master[n] for n in g["names"]], "lr": g["lr"], "weight_decay": g["weight_decay"]} for g in groups],
                                betas=(0.9, 0.95), eps=1e-8)
    total = torch.nn.utils.clip_grad_norm_([master[n] for n, _ in named], clip_at)
    assert float(total) > clip_at, "the test must actually clip"
    ref_opt.step()
    # fused path
    opt = OPT.FusedAdamW(model.engine, lr=lr, betas=(0.9, 0.95), eps=1e-8, weight_decay=wd, lr_scale_fn=OPT.vit_lr_scale)
    before = {n: p.detach().clone() for n, p in named}
    opt.step(max_grad_norm=clip_at, lr_mult=mult)
    assert abs(float(opt.last_grad_norm()) - float(total)) / float(total) < 2e-3
    worst = 0.0
    for n, p in named:
        ref = master[n].detach()
        got = p.detach().float()
        moved = (before[n].float() - ref).abs().max()
        err = (got - ref).abs().max()
        ulp = ref.abs().max() * 2.0 ** -8 + 1e-12          # one bf16 rounding at the tensor's scale
        assert float(err) <= float(ulp) * 1.01, (n, float(err), float(ulp))
        worst = max(worst, float(err))
    assert worst > 0.0  # (bf16 storage: the fp32 result is not reproduced exactly, only to rounding)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_ragged_batch_equals_per_sequence_runs_at_real_widths(dtype):
    """The ragged variant of cfg 3 (lengths 4096, 3900, ... exercise the key-padding branch,
    llama_flash_attn_monkey_patch.py:87-102) as a size-independent property at the real head width (d 4096, 32 heads of
    128, 2 layers): a right-padded batch must give, at every valid position, the logits of the same sequence run alone
    without padding, and the loss must be the token-weighted combination of the single-sequence losses; gradients of the
    padded run must not depend on what the pad positions hold."""
    from merlin_amd import synth
    from oracle import cases as C

    cfg = C.medium_cfg()
    S, lens = 768, [768, 701, 130]
    P_img = cfg.num_patches
    samples = []
    g = torch.Generator().manual_seed(3)
    for b, L_ in enumerate(lens):
        one = synth.interpair_batch(B=1, S=L_, frames=1, base_vocab=cfg.vocab_size - 3, P=P_img, image_size=cfg.v_image_size,
                                    rank=b) if L_ > P_img + 16 else None
        if one is None:  # a text-only row (with the reference's zeros image)
            ids = torch.randint(3, cfg.vocab_size - 3, (1, L_), generator=g)
            ids[0, 0] = 1
            lab = ids.clone(); lab[0, : L_ // 2] = -100
            one = dict(input_ids=ids, labels=lab, attention_mask=torch.ones(1, L_, dtype=torch.bool), images=[torch.zeros(1, 3, cfg.v_image_size, cfg.v_image_size)])
        samples.append(one)
    pad = lambda t, v: torch.cat([t, torch.full((1, S - t.shape[1]), v, dtype=t.dtype)], dim=1)
    batch = dict(input_ids=torch.cat([pad(s["input_ids"], 0) for s in samples]), labels=torch.cat([pad(s["labels"], -100) for s in samples]),
                 attention_mask=torch.cat([pad(s["attention_mask"].to(torch.bool), False) for s in samples]), images=[s["images"][0] for s in samples])
    model = _build(cfg, dtype)
    out = model(**_to_dev(batch))
    out.loss.backward()
    g_pad = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    for p in model.parameters():
        p.grad = None
    tol = TOL[dtype]["logits"]
    num, den = 0.0, 0
    for b, s in enumerate(samples):
        o1 = model(**_to_dev(s))
        L_ = lens[b]
        ref = o1.logits[0].float()
        got = out.logits[b, :L_].float()
        assert float((got - ref).abs().max() / ref.abs().max()) < tol, b
        n = int((s["labels"][0, 1:] != -100).sum())
        num += float(o1.loss) * n
        den += n
    assert abs(float(out.loss) - num / den) / (num / den) < TOL[dtype]["loss"]
    # pad contents are irrelevant: garbage ids in the pad region, same mask -> identical loss and gradients
    junk = dict(batch)
    ids2 = batch["input_ids"].clone()
    for b, L_ in enumerate(lens):
        ids2[b, L_:] = 7
    junk["input_ids"] = ids2
    out2 = model(**_to_dev(junk))
    out2.loss.backward()
    assert float(out2.loss) == float(out.loss)
    for k, p in model.named_parameters():
        if p.grad is not None:
            assert torch.equal(p.grad, g_pad[k]), k


def test_fp8_weight_decode_tracks_16bit_decode():
    """Decode with fp8 (e4m3, per-128-block scales) decoder weights against the 16-bit decode on the same cache: the
    logits that pick each token differ only by the weight quantisation (stated tolerance: 6e-2 of the logit range on the
    2-layer model; the prefill and the cache are identical)."""
    from oracle import cases as C

    cfg, batch = C.get_case("tiny_1img")
    model = _build(cfg, torch.bfloat16)
    b = _to_dev(batch)
    n_new = 5
    logits, cache = model.engine.prefill(b["input_ids"], b["attention_mask"], b["images"], n_new)
    logits8, cache8 = model.engine.prefill(b["input_ids"], b["attention_mask"], b["images"], n_new)
    assert torch.equal(logits, logits8)
    for _ in range(n_new - 1):
        nxt = logits.argmax(dim=-1)
        logits = model.engine.decode_step(nxt, cache)
        logits8 = model.engine.decode_step(nxt, cache8, fp8=True)   # teacher-forced on the 16-bit path's tokens
        err = float((logits8 - logits).abs().max() / logits.abs().max())
        assert err < 6e-2, err
    out = model.generate(b["input_ids"], images=b["images"], max_new_tokens=4, fp8_weights=True, eos_token_id=-1)
    assert out.shape == (1, b["input_ids"].shape[1] + 4)


@pytest.mark.parametrize("name", ["tiny_2img", "medium_cfg1"])
def test_fp8_forward_tracks_reference_golden(name):
    """Forward with every decoder Linear on the scaled-fp8 MFMA (e4m3 operands, per-token / per-channel scales) against the
    reference's fp32 logits: stated tolerance max|d| <= 0.15 and rms(d) <= 0.04 of the logit range, 2 % on the loss, on
    the 2-layer models (the 16-bit path holds 1.5e-2 / 5e-3; measured here 0.10 max) - the price of 3 mantissa bits on
    both operands; the fp8 path refuses to run with gradients."""
    from oracle import cases as C

    cfg, batch = C.get_case(name)
    g = np.load(os.path.join(GOLD, name + ".npz"))
    model = _build(cfg, torch.bfloat16)
    model.fp8_forward = True
    with torch.no_grad():
        out = model(**_to_dev(batch))
    lg = out.logits.float()
    if "logits_slice" in g.files:
        dlt, rng = lg[:, ::8, :512].cpu().numpy() - g["logits_slice"], float(g["logits_absmax"])
    else:
        dlt, rng = lg.cpu().numpy() - g["logits"], float(np.abs(g["logits"]).max())
    m = batch["attention_mask"].numpy().astype(bool)
    if dlt.shape[1] == m.shape[1]:
        dlt = dlt[m]
    assert np.abs(dlt).max() / rng < 0.15, np.abs(dlt).max() / rng
    assert np.sqrt((dlt ** 2).mean()) / rng < 0.04, np.sqrt((dlt ** 2).mean()) / rng
    assert abs(float(out.loss) - float(g["loss"])) < 2e-2 * abs(float(g["loss"]))
    with pytest.raises(RuntimeError):
        model(**_to_dev(batch)).loss
