"""The oracle's decoding loops (oracle/gen_ref.py: greedy / warpers / beam search restated from transformers' GenerationMixin)
against goldens captured in the build container (oracle/make_gen_golden.py):
  * greedy: token ids from the REAL reference's `model.generate` on multimodal prompts,
  * beam search: token ids from transformers' own generate on a plain LlamaForCausalLM carrying the same decoder weights,
  * warper chain: probabilities from transformers' Temperature/TopK/TopP warpers.
The GPU tests (tests/test_generation_gpu.py) then compare merlin_amd's generate() with the same goldens and oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cases as C
from oracle import gen_ref as G
from oracle import ref_cpu as R

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "gen_tiny.json")))


def _logits_fn(P, cfg, images):
    def fn(x):
        with torch.no_grad():
            imgs = None if images is None else (images * x.shape[0] if x.shape[0] > len(images) else images)
            _, lg = R.forward(P, cfg, x, None, None, imgs)
        return lg[:, -1, :]
    return fn


@pytest.mark.parametrize("i", range(len(GOLD["cases"])))
def test_oracle_greedy_matches_reference_generate(i):
    rec = GOLD["cases"][i]
    cfg, batch = C.get_case(rec["case"])
    P = R.make_params(cfg, seed=0)
    P["model.norm.weight"] = P["model.norm.weight"] * rec["logit_gain"]
    ids = batch["input_ids"][:, :rec["prompt_len"]]
    g = G.sample(_logits_fn(P, cfg, batch["images"]), ids, rec["max_new_tokens"], eos_ids=[rec["eos_token_id"]], pad=0)
    assert g.tolist() == rec["greedy"]


@pytest.mark.parametrize("i", range(len(GOLD["beam_cases"])))
def test_oracle_beam_search_matches_transformers(i):
    rec = GOLD["beam_cases"][i]
    cfg, _ = C.get_case("tiny_1img")
    P = R.make_params(cfg, seed=0)
    P["model.norm.weight"] = P["model.norm.weight"] * rec["logit_gain"]
    ids = torch.tensor(rec["prompt"], dtype=torch.int64)
    fn = _logits_fn(P, cfg, None)
    g = G.sample(fn, ids, rec["max_new_tokens"], eos_ids=[rec["eos_token_id"]], pad=0)
    assert g.tolist() == rec["greedy"]
    b = G.beam_search(fn, ids, rec["num_beams"], rec["max_new_tokens"], eos_ids=[rec["eos_token_id"]], pad=0, length_penalty=rec["length_penalty"])
    assert b.tolist() == rec["beam"]


def test_warper_chain_matches_transformers():
    for row in GOLD["warper_rows"]:
        got = G.warp_probs(np.array(row["logits"], dtype=np.float32), row["temperature"], row["top_k"], row["top_p"])
        ref = np.array(row["probs"])
        assert np.abs(got - ref).max() < 1e-6 and ((got > 0) == (ref > 0)).all()


def test_counter_uniform_is_reproducible_and_uniform():
    u = np.array([G.counter_uniform(123, s, r) for s in range(200) for r in range(5)])
    assert (u >= 0).all() and (u < 1).all() and abs(u.mean() - 0.5) < 0.03 and len(set(u.tolist())) > 990
    assert G.counter_uniform(123, 7, 2) == G.counter_uniform(123, 7, 2) != G.counter_uniform(124, 7, 2)


def test_generate_options_are_implemented_neutral_or_refused_by_name():
    """generate() must not swallow unknown keyword arguments (serve/cli.py passes streamer=..., HF callers pass
    return_dict_in_generate / generation_config / logits_processor ...)."""
    import pytest
    from merlin_amd.generation import _resolve_options

    o = _resolve_options(None, dict(do_sample=True, temperature=0.2, max_new_tokens=7, output_scores=False, return_dict_in_generate=False,
                                    logits_processor=None, min_length=0, repetition_penalty=1.0))
    assert o["do_sample"] is True and o["temperature"] == 0.2 and o["max_new_tokens"] == 7 and o["top_k"] == 50 and o["num_beams"] == 1
    for bad in (dict(return_dict_in_generate=True), dict(output_scores=True), dict(repetition_penalty=1.2), dict(min_length=5),
                dict(logits_processor=[object()]), dict(num_return_sequences=3)):
        with pytest.raises(NotImplementedError):
            _resolve_options(None, bad)
    with pytest.raises(TypeError):
        _resolve_options(None, dict(not_a_generate_option=1))

    class GC:  # a transformers.GenerationConfig stand-in: fields first, explicit kwargs over them
        num_beams, max_new_tokens, do_sample, temperature, repetition_penalty = 5, 12, False, None, 1.0
    o = _resolve_options(GC(), dict(max_new_tokens=3))
    assert o["num_beams"] == 5 and o["max_new_tokens"] == 3 and o["temperature"] == 1.0
    GC.repetition_penalty = 1.3
    with pytest.raises(NotImplementedError):
        _resolve_options(GC(), {})
