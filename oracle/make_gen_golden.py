"""Golden vectors for generate(): token ids produced by the REAL reference's `model.generate` (transformers GenerationMixin
driving mmgpt's MMGPTLlamaForCausalLM, CPU fp32) in the build container, for the decoding modes its eval scripts use
(eval_mmvet.py:101-120): greedy and num_beams=5 beam search (with `use_beam_search` set, as data_args does).  Also pins the
oracle's warper chain against transformers' own logits warpers on random rows.

    python oracle/make_gen_golden.py      -> tests/golden/gen_tiny.json

The tiny model's random-init logits are nearly flat; `logit_gain` multiplies model.norm.weight so that the next-token
distributions are peaked enough for beam search to differ from greedy decoding and for EOS to be reached."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases as C  # noqa: E402
from oracle import gen_ref as G  # noqa: E402
from oracle import make_golden as MG  # noqa: E402


def hf_llama(cfg, gain):
    """Plain transformers LlamaForCausalLM (the class the reference's model inherits `generate` from, llama_mmgpt.py:38) with
    the SAME generator weights as the oracle's decoder: the text-only form of the reference model, whose KV cache works
    under the installed transformers."""
    from transformers import LlamaConfig, LlamaForCausalLM

    from merlin_amd import weights as W

    hf = LlamaConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                     num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                     num_key_value_heads=cfg.num_attention_heads, rms_norm_eps=cfg.rms_norm_eps, max_position_embeddings=8192,
                     rope_theta=cfg.rope_theta, tie_word_embeddings=False, attn_implementation="eager")
    m = LlamaForCausalLM(hf).float().eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(torch.from_numpy(W.generate(n, tuple(p.shape), 0)))
        m.model.norm.weight.mul_(gain)
    return m


def main():
    out = {"cases": [], "beam_cases": []}
    # ---- greedy: the REAL reference (multimodal prompt, images spliced) ----
    for case, gain, eos, n_prompt, max_new in (("tiny_1img", 40.0, 37, 22, 12), ("tiny_1img", 25.0, 36, 22, 14), ("tiny_1img", 25.0, 54, 22, 14),
                                                 ("tiny_2img", 25.0, 3, 48, 14), ("tiny_1img", 25.0, 96, 22, 16)):
        cfg, batch = C.get_case(case)
        model = MG.build_reference_model(cfg, seed=0)
        model.eval()
        with torch.no_grad():
            model.model.norm.weight.mul_(gain)
        ids = batch["input_ids"][:, :n_prompt]
        rec = {"case": case, "logit_gain": gain, "eos_token_id": eos, "prompt_len": n_prompt, "max_new_tokens": max_new}
        with torch.no_grad():
            g = model.generate(ids, images=batch["images"], max_new_tokens=max_new, do_sample=False, eos_token_id=eos, pad_token_id=0)
        rec["greedy"] = g.tolist()
        print(case, gain, eos, "greedy", g[0, n_prompt:].tolist(), flush=True)
        out["cases"].append(rec)
    # ---- beam search: transformers' own generate on the decoder (text-only prompts; see module docstring of gen_ref.py:
    # the reference's multimodal beam path needs transformers 4.31's tuple KV cache - under the installed version its
    # repeat_interleave(5) hack runs cache-less and feeds beam 0's tokens to every beam, so it cannot serve as a pin) ----
    cfg, _ = C.get_case("tiny_1img")
    rng = np.random.RandomState(7)
    for gain, eos, n_prompt, max_new, nb, lp, B in ((25.0, 24, 9, 14, 5, 1.0, 1), (25.0, 31, 12, 14, 5, 1.0, 1), (40.0, 57, 7, 12, 5, 1.0, 2),
                                                      (15.0, 94, 10, 16, 5, 1.0, 1), (25.0, 92, 8, 12, 3, 2.0, 2), (25.0, 52, 8, 12, 5, 0.0, 1),
                                                      (30.0, 46, 6, 10, 5, 1.0, 1), (10.0, 61, 11, 12, 4, 1.0, 3), (15.0, 99, 10, 16, 5, 1.0, 1)):
        m = hf_llama(cfg, gain)
        ids = torch.from_numpy(rng.randint(3, 100, size=(B, n_prompt)).astype(np.int64))
        ids[:, 0] = 1
        with torch.no_grad():
            g = m.generate(ids, max_new_tokens=max_new, do_sample=False, eos_token_id=eos, pad_token_id=0)
            b = m.generate(ids, max_new_tokens=max_new, num_beams=nb, length_penalty=lp, eos_token_id=eos, pad_token_id=0)
        out["beam_cases"].append({"logit_gain": gain, "eos_token_id": eos, "prompt": ids.tolist(), "max_new_tokens": max_new, "num_beams": nb,
                                  "length_penalty": lp, "greedy": g.tolist(), "beam": b.tolist()})
        print("beam", gain, eos, nb, lp, [r[n_prompt:] for r in g.tolist()], [r[n_prompt:] for r in b.tolist()], flush=True)
    # warper chain vs transformers' own processors
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper

    rng = np.random.RandomState(5)
    rows = []
    for (T, k, p) in ((0.2, 50, 1.0), (1.0, 5, 1.0), (0.7, 50, 0.9), (1.3, 0, 0.5)):
        x = (rng.standard_normal(203) * 3).astype(np.float32)
        s = torch.from_numpy(x)[None].clone()
        s = TemperatureLogitsWarper(T)(None, s)
        if k:
            s = TopKLogitsWarper(k)(None, s)
        if p < 1:
            s = TopPLogitsWarper(p)(None, s)
        ref = torch.softmax(s, -1)[0].numpy().astype(np.float64)
        got = G.warp_probs(x, T, k, p)
        assert np.abs(ref - got).max() < 1e-6 and ((ref > 0) == (got > 0)).all(), (T, k, p)
        rows.append({"temperature": T, "top_k": k, "top_p": p, "logits": x.tolist(), "probs": ref.tolist()})
    out["warper_rows"] = rows
    path = os.path.join(ROOT, "tests", "golden", "gen_tiny.json")
    json.dump(out, open(path, "w"))
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
