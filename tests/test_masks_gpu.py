"""Arbitrary key-padding masks (llama_flash_attn_monkey_patch.py:87-102): the reference's flash path honours ANY
attention_mask through unpad_input / pad_input, and batched `generate` with HF's default LEFT padding relies on it.  The HIP path
keeps the lens-only fast path for right-padded batches and routes every other mask through the same unpad -> varlen causal attention
-> pad sequence (mh_mask_unpad_index + mh_gather_rows2d).  Checked against the fp32 CPU oracle on the valid rows, against the fast
path (bit-identical on right-padded input), and in generate() against un-padded single-prompt decoding."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_unpad_index_tables_and_zero_row_gather():
    from merlin_amd import ops as O

    g = torch.Generator().manual_seed(0)
    for B, S in ((1, 5), (3, 64), (4, 613), (2, 8192)):
        m = torch.rand(B, S, generator=g) > 0.35
        m[0] = True
        if B > 1:
            m[1, : S // 2] = False  # left padding
        if B > 2:
            m[2] = False            # an empty row
        fwd, inv, cnt = O.mask_unpad_index(m.cuda().contiguous())
        fwd, inv, cnt = fwd.cpu().view(B, S), inv.cpu().view(B, S), cnt.cpu()
        for b in range(B):
            pos = m[b].nonzero().view(-1)
            n = int(pos.numel())
            assert int(cnt[b]) == n
            assert torch.equal(fwd[b, :n], b * S + pos) and bool((fwd[b, n:] == -1).all())
            want = torch.full((S,), -1, dtype=torch.int64)
            want[pos] = b * S + torch.arange(n)
            assert torch.equal(inv[b], want)
    src = torch.randn(40, 24, device="cuda").to(torch.float16)
    idx = torch.tensor([3, -1, 39, 0, -1], device="cuda")
    dst = torch.full((5, 24), 7.0, device="cuda", dtype=torch.float16)
    O.gather_rows2d(src, idx, dst)
    assert torch.equal(dst[[0, 2, 3]], src[[3, 39, 0]]) and float(dst[[1, 4]].abs().max()) == 0.0


def _masked_case(kind):
    from oracle import cases as C

    cfg, batch = C.get_case("tiny_padbatch")
    ids, am, labels = batch["input_ids"].clone(), batch["attention_mask"].clone().bool(), batch["labels"].clone()
    B, S = ids.shape
    if kind == "left":  # every row's valid tokens moved to the END (HF left padding), images stay inside the valid part
        for b in range(B):
            n = int(am[b].sum())
            ids[b] = torch.cat([ids[b, n:], ids[b, :n]])
            labels[b] = torch.cat([labels[b, n:], labels[b, :n]])
            am[b] = torch.cat([am[b, n:], am[b, :n]])
            labels[b, S - n] = -100  # the first valid token would be predicted from a pad row (garbage in both implementations)
    elif kind == "holes":  # text positions knocked out in the middle of the valid part (never an image token)
        rng = np.random.RandomState(3)
        for b in range(B):
            n = int(am[b].sum())
            text = [s for s in range(1, n) if int(ids[b, s]) < cfg.vocab_size - 3]
            for s in rng.choice(text, size=max(1, len(text) // 4), replace=False):
                am[b, s] = False
                # a masked-out token still exists as a QUERY row in the reference's eager CPU path but gets a zero attention output in
                # its flash path (pad_input): its own logits are excluded from every comparison, so nothing may be predicted from it
                if s + 1 < S:
                    labels[b, s + 1] = -100
    return cfg, dict(input_ids=ids, attention_mask=am, labels=labels, images=batch["images"])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("kind", ["left", "holes"])
def test_general_mask_forward_backward_vs_oracle(kind, dtype):
    from oracle import ref_cpu as R
    from test_model_gpu import TOL, _build, _to_dev

    cfg, batch = _masked_case(kind)
    model = _build(cfg, dtype)
    out = model(**_to_dev(batch))
    out.loss.backward()
    P = R.make_params(cfg, seed=0, requires_grad=True)
    loss_ref, logits_ref = R.forward(P, cfg, batch["input_ids"], batch["attention_mask"], batch["labels"], batch["images"])
    loss_ref.backward()
    valid = batch["attention_mask"].bool()
    got, want = out.logits.float().cpu()[valid], logits_ref.detach()[valid]  # padded rows differ by construction (§8: compare valid rows)
    err = float((got - want).abs().max() / want.abs().max())
    tol = 1.5e-3 if dtype == torch.float16 else 2e-2
    assert err < tol, err
    assert abs(float(out.loss) - float(loss_ref)) < (3e-3 if dtype == torch.float16 else 3e-2) * abs(float(loss_ref))
    named = dict(model.named_parameters())
    bad = []
    for k in ("model.layers.0.self_attn.q_proj.weight", "model.layers.0.self_attn.k_proj.weight", "model.layers.1.self_attn.v_proj.weight",
              "model.layers.0.mlp.down_proj.weight", "lm_head.weight", "model.projector.projector.weight"):
        g, gr = named[k].grad.float().cpu().reshape(-1), P[k].grad.reshape(-1)
        cos = float(g @ gr / (g.norm() * gr.norm() + 1e-30))
        if cos < (0.999 if dtype == torch.float16 else 0.99):
            bad.append((k, cos))
    assert not bad, bad
    del TOL


def test_right_padded_batch_is_bit_identical_through_the_general_path():
    """The unpad / pad route on a right-padded batch = the lens-only fast path: bit for bit in the forward (same kernels on the same
    rows); in the backward up to one extra 16-bit rounding of dq / dk (the inverse RoPE runs on the un-packed rows as a separate
    pass instead of in the attention kernels' fp32 epilogue); also with layer recompute (packed copies rebuilt in the backward)."""
    from oracle import cases as C
    from test_model_gpu import _build, _to_dev

    cfg, batch = C.get_case("tiny_padbatch")
    b = _to_dev(batch)
    model = _build(cfg, torch.bfloat16)
    valid = b["attention_mask"].bool()
    for save in (True, False):  # (the tower's fc1 + quick-GELU is one fused launch without resident activations: compare like with like)
        model.engine.save_activations = save
        res = []
        for general in (False, True):
            for p in model.parameters():
                p.grad = None
            model.engine.force_unpad = general
            out = model(**b)
            out.loss.backward()
            res.append((out.logits[valid].clone(), out.loss.clone(), model.engine.arena.gflat.clone()))
        (l0, loss0, g0), (l1, loss1, g1) = res
        assert torch.equal(l1, l0) and torch.equal(loss1, loss0)
        cos = float((g1.float() * g0.float()).sum() / (g1.float().norm() * g0.float().norm()))
        assert cos > 0.99995 and float((g1.float() - g0.float()).abs().max()) <= 0.02 * float(g0.float().abs().max()), cos
    model.engine.force_unpad = False


def test_left_padded_batched_generate_matches_unpadded_prompts():
    """HF's default for batched decoding is LEFT padding.  RoPE scores depend on position differences only, so each row of the
    left-padded batch must produce the tokens of its own un-padded prompt (ties within fp16 rounding aside); the streamer protocol
    (serve/cli.py passes a TextStreamer) receives the prompt, every step's tokens, then end()."""
    from test_generation_gpu import GOLD, _model, _same_or_tie

    rec = GOLD["cases"][0]
    cfg, batch, m = _model("tiny_2img", rec["logit_gain"])
    full = batch["input_ids"]
    n_img = [int(((full[b] == cfg.im_start_token)).sum()) for b in range(full.shape[0])]
    prompts, imgs = [], []
    for b, cut in ((0, full.shape[1]), (0, full.shape[1] - 5)):  # two prompts of different lengths over the same images
        prompts.append(full[b, :cut])
        imgs.append(batch["images"][b])
    assert n_img[0] > 0
    P = max(int(p.numel()) for p in prompts)
    ids = torch.zeros(len(prompts), P, dtype=torch.int64)
    am = torch.zeros(len(prompts), P, dtype=torch.bool)
    for i, p in enumerate(prompts):
        ids[i, P - p.numel():] = p
        am[i, P - p.numel():] = True
    images = [im.cuda() for im in imgs]
    kw = dict(max_new_tokens=8, do_sample=False, eos_token_id=rec["eos_token_id"], pad_token_id=0)

    class Rec:
        def __init__(self):
            self.put_calls, self.ended = [], False

        def put(self, v):
            self.put_calls.append(v.clone())

        def end(self):
            self.ended = True

    st = Rec()
    got = m.generate(ids.cuda(), images=images, attention_mask=am.cuda(), streamer=st, **kw).cpu()
    assert got.shape[0] == 2 and torch.equal(got[:, :P], ids)  # HF layout: continuation appended after the padded prompt
    assert st.ended and torch.equal(st.put_calls[0], ids) and len(st.put_calls) == 1 + (got.shape[1] - P)
    assert torch.equal(torch.stack(st.put_calls[1:], 1), got[:, P:])
    for i, p in enumerate(prompts):
        want = m.generate(p[None].cuda(), images=images[i:i + 1], **kw).cpu()
        n = min(want.shape[1] - p.numel(), got.shape[1] - P)
        _same_or_tie(m, images[i:i + 1], torch.cat([p[None], got[i:i + 1, P:P + n]], 1), want[:, :p.numel() + n], int(p.numel()), ("row", i))
    # without a cache (full recompute per token through the unpad path) the same tokens come out
    got_nc = m.generate(ids.cuda(), images=images, attention_mask=am.cuda(), use_cache=False, **kw).cpu()
    n = min(got.shape[1], got_nc.shape[1])
    assert float((got[:, :n] != got_nc[:, :n]).float().mean()) < 0.2
    with pytest.raises(TypeError):
        m.generate(ids.cuda(), images=images, attention_mask=am.cuda(), definitely_not_an_option=1)
    with pytest.raises(NotImplementedError):
        m.generate(ids.cuda(), images=images, attention_mask=am.cuda(), return_dict_in_generate=True)


def test_mask_classification_does_not_depend_on_strict_checks_and_generate_follows_the_engine():
    """with engine.strict_checks = False a left-padded batch used to be treated as dense (pads became valid keys) while
    generate() classified the mask on its own.  The classification is now always made on the device, and generate() follows the
    engine's decision (cache.rpos): strict on / off give bit-identical logits on a left-padded batch, and a right-padded batch sent
    through engine.force_unpad generates the same tokens as through the lens fast path."""
    from oracle import cases as C
    from test_generation_gpu import GOLD, _model
    from test_model_gpu import _build, _to_dev

    cfg, batch = C.get_case("tiny_padbatch")
    b = _to_dev(batch)
    model = _build(cfg, torch.float16)
    am = b["attention_mask"].bool()
    lens = am.sum(1)
    S = am.shape[1]
    # left-pad: move every row's valid tokens to the end
    ids_l, lab_l, am_l = torch.zeros_like(b["input_ids"]), torch.full_like(b["labels"], -100), torch.zeros_like(am)
    for i in range(am.shape[0]):
        n = int(lens[i])
        ids_l[i, S - n:], lab_l[i, S - n:], am_l[i, S - n:] = b["input_ids"][i, :n], b["labels"][i, :n], True
    res = []
    with torch.no_grad():
        for strict in (True, False):
            model.engine.strict_checks = strict
            out = model(input_ids=ids_l, attention_mask=am_l, labels=lab_l, images=b["images"])
            res.append((out.logits[am_l].clone(), out.loss.clone()))
    model.engine.strict_checks = True
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    if int(lens.min()) < S:  # the batch really has padding: a dense reading would differ
        with torch.no_grad():
            dense = model(input_ids=ids_l, attention_mask=None, labels=lab_l, images=b["images"]).logits[am_l]
        assert not torch.equal(dense, res[0][0])
    # generate(): right-padded prompts through the fast path and through force_unpad
    rec = GOLD["cases"][0]
    cfg2, batch2, m = _model("tiny_padbatch", rec["logit_gain"])
    kw = dict(max_new_tokens=6, do_sample=False, eos_token_id=rec["eos_token_id"], pad_token_id=0)
    ids2, am2 = batch2["input_ids"].cuda(), batch2["attention_mask"].cuda()
    images2 = [im.cuda() for im in batch2["images"]]
    fast = m.generate(ids2, images=images2, attention_mask=am2, **kw).cpu()
    m.engine.force_unpad = True
    try:
        general = m.generate(ids2, images=images2, attention_mask=am2, **kw).cpu()
    finally:
        m.engine.force_unpad = False
    # fast path: continuation compacted behind each row's own length; general path (HF layout): appended after the padded prompt
    P = ids2.shape[1]
    l2 = am2.sum(1).cpu()
    agree = 0
    for i in range(ids2.shape[0]):
        n = min(fast.shape[1] - int(l2[i]), general.shape[1] - P)
        agree += int((fast[i, int(l2[i]):int(l2[i]) + n] == general[i, P:P + n]).sum())
    assert agree >= 0.8 * 6 * ids2.shape[0], (fast, general)  # (RoPE position offsets differ between the two layouts: fp16 ties aside)
