"""generate() on the HIP path (SURVEY §8f N3, §8b "Eval calls model.generate(...)") against
  * token ids produced by the REAL reference's `model.generate` (greedy, multimodal prompts) and by transformers' own beam
    search on the same decoder weights (tests/golden/gen_tiny.json, oracle/make_gen_golden.py),
  * the CPU oracle's decoding loops (oracle/gen_ref.py, pinned to those goldens in tests/test_generation_cpu.py) driven by the
    fp32 oracle forward: sampling with a fixed seed, beam search on multimodal prompts, stopping criteria,
and the selection kernels against numpy / torch restatements of transformers' logits warpers."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "gen_tiny.json")))


def _model(case, gain, dtype=torch.float16):
    from oracle import cases as C
    from test_model_gpu import _build

    cfg, batch = C.get_case(case)
    m = _build(cfg, dtype)
    with torch.no_grad():
        m.model.norm.weight.mul_(gain)  # the goldens' peaked-logits variant of the tiny model
    m.engine.weights_changed()
    return cfg, batch, m


def _oracle_fn(cfg, gain, images):
    from oracle import ref_cpu as R

    P = R.make_params(cfg, seed=0)
    P["model.norm.weight"] = P["model.norm.weight"] * gain

    def fn(x):
        with torch.no_grad():
            imgs = None if images is None else (images * x.shape[0] if x.shape[0] > len(images) else images)
            return R.forward(P, cfg, x, None, None, imgs)[1][:, -1, :]
    return fn


# ---- kernels ---------------------------------------------------------------------------------------------------------
def test_select_tokens_greedy_and_warpers_vs_transformers_rows():
    from merlin_amd import ops as O
    from oracle import gen_ref as G

    g = torch.Generator().manual_seed(3)
    x = (torch.randn(7, 32064, generator=g) * 2).cuda()
    V = 32003
    assert torch.equal(O.select_tokens(x, V), x[:, :V].argmax(-1))
    x[2, 5] = x[2, 77] = 50.0  # tie: lowest index wins
    assert int(O.select_tokens(x, V)[2]) == 5
    # sampling: the chosen token must be the inverse-CDF pick of HF's warped distribution at the kernel's own uniform
    for row in GOLD["warper_rows"]:
        lg = torch.tensor(row["logits"], dtype=torch.float32).cuda()[None].repeat(64, 1).contiguous()
        probs = np.array(row["probs"])
        cdf = np.cumsum(probs)
        for step in range(4):
            tok, u = O.select_tokens(lg, do_sample=True, temperature=row["temperature"], top_k=row["top_k"], top_p=row["top_p"], seed=99, step=step, return_u=True)
            tok, u = tok.cpu().numpy(), u.cpu().numpy()
            for r in range(64):
                assert abs(u[r] - G.counter_uniform(99, step, r)) < 1e-7
                t = int(tok[r])
                assert probs[t] > 0, "sampled a token the warpers removed"
                lo = cdf[t - 1] if t > 0 else 0.0
                assert lo - 2e-5 <= u[r] <= cdf[t] + 2e-5, (row["temperature"], row["top_k"], row["top_p"], r, t, u[r], lo, cdf[t])


def test_select_tokens_distribution():
    from merlin_amd import ops as O
    from oracle import gen_ref as G

    rng = np.random.RandomState(0)
    x = (rng.standard_normal(40) * 2).astype(np.float32)
    probs = G.warp_probs(x, 0.8, 10, 0.95)
    lg = torch.from_numpy(x).cuda()[None].repeat(256, 1).contiguous()
    cnt = np.zeros(40)
    for step in range(40):
        t = O.select_tokens(lg, do_sample=True, temperature=0.8, top_k=10, top_p=0.95, seed=7, step=step).cpu().numpy()
        cnt += np.bincount(t, minlength=40)
    freq = cnt / cnt.sum()
    assert (freq[probs == 0] == 0).all()
    assert np.abs(freq - probs).max() < 0.02, np.abs(freq - probs).max()


def test_log_softmax_and_gather_rows():
    from merlin_amd import ops as O

    x = torch.randn(10, 32064, device="cuda") * 3
    bias = torch.randn(10, device="cuda")
    ref = torch.log_softmax(x[:, :32003], -1) + bias[:, None]
    assert float((O.log_softmax_rows(x, 32003, row_bias=bias) - ref).abs().max()) < 1e-4
    src = torch.randn(6, 40, 64, device="cuda", dtype=torch.bfloat16)
    idx = torch.tensor([3, 3, 0, 5, 1, 1, 2], device="cuda")
    dst = torch.zeros(7, 40, 64, device="cuda", dtype=torch.bfloat16)
    O.gather_rows2d(src.view(6, -1), idx, dst.view(7, -1), cols=17 * 64)
    assert torch.equal(dst[:, :17], src[idx][:, :17]) and float(dst[:, 17:].abs().max()) == 0.0


def _same_or_tie(m, images, got, want, n_prompt, tag=None):
    """Token-for-token equality, except that a decoding step whose two best candidates are within fp16 rounding of each other
    (checked on the HIP logits of the golden prefix) may legitimately pick the other one; everything before it must match."""
    if got.tolist() == want.tolist():
        return
    n = min(got.shape[1], want.shape[1])
    t = int((got[:, :n] != want[:, :n]).any(0).nonzero()[0]) if bool((got[:, :n] != want[:, :n]).any()) else n
    assert t >= n_prompt, (tag, got.tolist(), want.tolist())
    b = int((got[:, t] != want[:, t]).nonzero()[0])
    with torch.no_grad():
        lg = m(input_ids=want[b:b + 1, :t].cuda(), images=images[b:b + 1] if images is not None else None).logits[0, -1].float()
    gap = float(lg.max() - lg[int(want[b, t])])
    assert gap < 2e-3 * float(lg.abs().max()), (tag, "diverged at", t, "without a tie", gap, got.tolist(), want.tolist())


# ---- greedy: the REAL reference's generate ------------------------------------------------------------------------------
@pytest.mark.parametrize("i", range(len(GOLD["cases"])))
def test_greedy_generate_matches_reference(i):
    rec = GOLD["cases"][i]
    cfg, batch, m = _model(rec["case"], rec["logit_gain"])
    ids = batch["input_ids"][:, :rec["prompt_len"]].cuda()
    images = [im.cuda() for im in batch["images"]]
    kw = dict(max_new_tokens=rec["max_new_tokens"], do_sample=False, eos_token_id=rec["eos_token_id"], pad_token_id=0)
    want = torch.tensor(rec["greedy"])
    for extra in (dict(), dict(use_graph=False), dict(use_cache=False)):
        got = m.generate(ids, images=images, **kw, **extra).cpu()
        _same_or_tie(m, images, got, want, rec["prompt_len"], extra)


# ---- beam search: transformers' own generate on the same decoder weights (text-only prompts) ----------------------------
@pytest.mark.parametrize("i", range(len(GOLD["beam_cases"])))
def test_beam_search_matches_transformers(i):
    rec = GOLD["beam_cases"][i]
    cfg, batch, m = _model("tiny_1img", rec["logit_gain"])
    ids = torch.tensor(rec["prompt"], dtype=torch.int64).cuda()
    got = m.generate(ids, max_new_tokens=rec["max_new_tokens"], num_beams=rec["num_beams"], length_penalty=rec["length_penalty"],
                     eos_token_id=rec["eos_token_id"], pad_token_id=0, temperature=0.2).cpu()
    assert got.tolist() == rec["beam"], (got.tolist(), rec["beam"])
    g = m.generate(ids, max_new_tokens=rec["max_new_tokens"], eos_token_id=rec["eos_token_id"], pad_token_id=0).cpu()
    _same_or_tie(m, None, g, torch.tensor(rec["greedy"]), ids.shape[1])


def test_eval_style_calls_multimodal_beam_and_sampling_vs_oracle():
    """eval_mmvet.py:101-120's two calls on a multimodal prompt: `num_beams=5, temperature=0.2, stopping_criteria=[...]` with
    use_beam_search set (base_mmgpt.py:162-163) and `do_sample=True, temperature=0.2, stopping_criteria=[...]`, against the
    oracle's decoding loops over the fp32 oracle forward (same counter-based sampling stream)."""
    from oracle import gen_ref as G

    gain, eos = 25.0, 96
    cfg, batch, m = _model("tiny_1img", gain)
    fn = _oracle_fn(cfg, gain, batch["images"])
    ids_cpu = batch["input_ids"][:, :22]
    ids = ids_cpu.cuda()
    images = [im.cuda() for im in batch["images"]]
    calls = []

    class Keyword:  # the reference's KeywordsStoppingCriteria protocol (mm_utils.py:62-85): __call__(output_ids, scores) -> bool
        def __init__(self, tok, start_len):
            self.tok, self.start_len = tok, start_len

        def __call__(self, output_ids, scores, **kw):
            calls.append(tuple(output_ids.shape))
            return bool((output_ids[0, self.start_len:] == self.tok).any())

    # beam search
    m.use_beam_search = True
    got = m.generate(ids, images=images, num_beams=5, temperature=0.2, max_new_tokens=16, eos_token_id=eos, pad_token_id=0,
                     stopping_criteria=[Keyword(-5, 22)]).cpu()
    # the module surface under use_beam_search: HF-expanded input_ids, one image entry -> rows cut to len(images), logits x5
    out5 = m(input_ids=ids.repeat(5, 1), images=images)
    out1 = m(input_ids=ids, images=images)
    m.use_beam_search = False
    assert out5.logits.shape[0] == 5 and torch.equal(out5.logits[3], m(input_ids=ids, images=images).logits[0]) and torch.equal(out5.logits[0], out1.logits[0])
    want = G.beam_search(fn, ids_cpu, 5, 16, eos_ids=[eos], pad=0)
    assert got.tolist() == want.tolist(), (got.tolist(), want.tolist())
    assert calls and calls[0] == (10, 23)  # criteria see the (1 + n_eos) * num_beams candidates, like StoppingCriteriaList
    # sampling, fixed seed; a keyword criterion that fires on a token of the sampled continuation stops the generation there
    want = G.sample(fn, ids_cpu, 12, eos_ids=[eos], pad=0, do_sample=True, temperature=0.2, top_k=50, seed=4242)
    got = m.generate(ids, images=images, do_sample=True, temperature=0.2, max_new_tokens=12, eos_token_id=eos, pad_token_id=0, seed=4242).cpu()
    assert got.tolist() == want.tolist(), (got.tolist(), want.tolist())
    new = want[0, 22:].tolist()
    kw_tok = new[min(3, len(new) - 1)]
    cut = new.index(kw_tok) + 1
    got = m.generate(ids, images=images, do_sample=True, temperature=0.2, max_new_tokens=12, eos_token_id=eos, pad_token_id=0, seed=4242,
                     stopping_criteria=[Keyword(kw_tok, 22)]).cpu()
    assert got[0, 22:].tolist() == new[:cut]
    # higher temperature: different seeds give different continuations, the same seed the same one
    a = m.generate(ids, images=images, do_sample=True, temperature=1.5, max_new_tokens=10, eos_token_id=-1, seed=1)
    b = m.generate(ids, images=images, do_sample=True, temperature=1.5, max_new_tokens=10, eos_token_id=-1, seed=2)
    a2 = m.generate(ids, images=images, do_sample=True, temperature=1.5, max_new_tokens=10, eos_token_id=-1, seed=1)
    assert torch.equal(a, a2) and not torch.equal(a, b)
