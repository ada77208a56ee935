"""generate() for MMGPTLlamaForCausalLM on the HIP engine.

The reference inherits transformers' `GenerationMixin.generate` (LlamaForCausalLM base, llama_mmgpt.py:38) and its eval
scripts drive it as (eval_mmvet.py:101-120, eval_box.py, demo.py, model_worker.py):

    model.generate(input_ids, images=[Tensor[n,3,H,W]], do_sample=True, temperature=0.2, max_new_tokens=1024,
                   stopping_criteria=[KeywordsStoppingCriteria(...)])
    model.generate(input_ids, images=..., num_beams=5, temperature=0.2, max_new_tokens=1024, stopping_criteria=[...])

This module restates the three decoding modes those calls reach - greedy, multinomial sampling with the HF logits warpers
(temperature -> top-k, HF's generation default top_k=50 -> top-p) and beam search (transformers/generation/utils.py `_sample`,
`_beam_search`, the vectorised formulation of the installed transformers) - on top of the engine's prefill + KV-cache decode
step.  Pins (tests/golden/gen_tiny.json, oracle/make_gen_golden.py): greedy = token ids from the REAL reference's generate on
multimodal prompts; beam search = transformers' own generate on a plain LlamaForCausalLM carrying the same decoder weights (the
reference's multimodal beam path needs transformers 4.31's tuple KV cache and cannot run correctly in the build container);
the warper chain = transformers' Temperature/TopK/TopP warpers.  Device work is in kernels (decode step, mh_select_tokens, mh_log_softmax_rows, mh_gather_rows2d);
what stays here is the control flow HF also keeps on the host (beam bookkeeping on [B, 2*num_beams] tensors, the stopping
criteria protocol: user criteria are called with (input_ids, scores) every step exactly like StoppingCriteriaList does).
"""
from __future__ import annotations

import torch

from . import ops as O

NEG = -1.0e9


def _as_list(x):
    if x is None:
        return []
    if isinstance(x, (list, tuple)):
        return list(x)
    if isinstance(x, torch.Tensor):
        return [int(v) for v in x.reshape(-1).tolist()]
    return [int(x)]


class _Stopper:
    """StoppingCriteriaList semantics: MaxLengthCriteria + EosTokenCriteria + the caller's criteria, OR-ed per sequence.
    User criteria may return a bool (old API, e.g. the reference's KeywordsStoppingCriteria, mm_utils.py:62-85: one verdict
    for the whole batch) or a bool tensor [n]."""

    def __init__(self, max_length, eos_ids, user):
        self.max_length, self.eos_ids, self.user = max_length, eos_ids, list(user or [])

    def __call__(self, ids, scores=None):
        n = ids.shape[0]
        done = torch.full((n,), ids.shape[1] >= self.max_length, dtype=torch.bool, device=ids.device)
        last = ids[:, -1]
        for e in self.eos_ids:
            done = done | (last == e)
        for c in self.user:
            r = c(ids, scores)
            r = torch.as_tensor(r, device=ids.device)
            done = done | (r.to(torch.bool) if r.dim() else r.to(torch.bool).expand(n))
        return done


def _resolve_lengths(prompt_len, max_new_tokens, max_length):
    if max_new_tokens is not None:
        return prompt_len + int(max_new_tokens)
    if max_length is not None:
        return int(max_length)
    return max(prompt_len + 1, 20)  # HF GenerationConfig default max_length = 20


# decoding options this module implements, with transformers' GenerationConfig defaults
_DEFAULTS = dict(max_new_tokens=None, max_length=None, eos_token_id=None, pad_token_id=None, do_sample=False, temperature=1.0, top_k=50,
                 top_p=1.0, num_beams=1, length_penalty=1.0, early_stopping=False, use_cache=True)
# options that are accepted only at the value that leaves the implemented modes unchanged (anything else is a decoding mode the
# reference's scripts never reach: refuse it instead of silently returning something different)
_NEUTRAL = dict(repetition_penalty=(1.0, None), no_repeat_ngram_size=(0, None), num_return_sequences=(1, None), num_beam_groups=(1, None),
                penalty_alpha=(None, 0.0), typical_p=(1.0, None), min_new_tokens=(None, 0), min_length=(0, None), bad_words_ids=(None,),
                return_dict_in_generate=(False, None), output_scores=(False, None), output_logits=(False, None),
                output_attentions=(False, None), output_hidden_states=(False, None), logits_processor=(None,), prefix_allowed_tokens_fn=(None,),
                encoder_no_repeat_ngram_size=(0, None), diversity_penalty=(0.0, None), epsilon_cutoff=(0.0, None), eta_cutoff=(0.0, None),
                min_p=(None,), renormalize_logits=(False, None), forced_bos_token_id=(None,), forced_eos_token_id=(None,),
                suppress_tokens=(None,), begin_suppress_tokens=(None,), assistant_model=(None,), synced_gpus=(False, None),
                bos_token_id=None, use_beam_search=None, negative_prompt_ids=(None,), num_assistant_tokens=None, trust_remote_code=None)


def _resolve_options(generation_config, kw):
    """HF precedence: GenerationConfig fields first, explicit keyword arguments over them; unknown names raise TypeError."""
    opt = dict(_DEFAULTS)
    if generation_config is not None:
        for k in list(_DEFAULTS) + list(_NEUTRAL):
            v = getattr(generation_config, k, None)
            if v is None:
                continue
            if k in _DEFAULTS:
                opt[k] = v
            elif _NEUTRAL[k] is not None and v not in _NEUTRAL[k] and v != []:
                raise NotImplementedError(f"generation_config.{k}={v!r} is not part of the decoding modes the reference's eval scripts use")
    for k, v in kw.items():
        if k in _DEFAULTS:
            opt[k] = _DEFAULTS[k] if v is None and k not in ("max_new_tokens", "max_length", "eos_token_id", "pad_token_id") else v
        elif k in _NEUTRAL:
            ok = _NEUTRAL[k]
            if ok is not None and v not in ok and v != []:
                raise NotImplementedError(f"generate({k}={v!r}) is not part of the decoding modes the reference's eval scripts use")
        else:
            raise TypeError(f"generate() got an unexpected keyword argument '{k}'")
    return opt


@torch.no_grad()
def generate(model, input_ids, images=None, attention_mask=None, generation_config=None, stopping_criteria=None, streamer=None,
             use_graph=True, fp8_weights=False, seed=None, **kw):
    """transformers' `GenerationMixin.generate` for the modes the reference reaches (module docstring).  `streamer` follows HF's
    protocol (`put(prompt ids)`, `put(next tokens)` every step, `end()`): serve/cli.py:93-104 passes a TextStreamer.  Every
    other HF option is either implemented, accepted at its neutral value, or refused by name - nothing is silently dropped."""
    o = _resolve_options(generation_config, kw)
    max_new_tokens, max_length, eos_token_id, pad_token_id = o["max_new_tokens"], o["max_length"], o["eos_token_id"], o["pad_token_id"]
    do_sample, temperature, top_k, top_p = o["do_sample"], o["temperature"], o["top_k"], o["top_p"]
    num_beams, length_penalty, early_stopping, use_cache = o["num_beams"], o["length_penalty"], o["early_stopping"], o["use_cache"]
    cfg = model.config
    eos_ids = _as_list(cfg.eos_token_id if eos_token_id is None else eos_token_id)
    pad = pad_token_id if pad_token_id is not None else (cfg.pad_token_id if cfg.pad_token_id is not None else (eos_ids[0] if eos_ids else 0))
    B, P = input_ids.shape
    max_len = _resolve_lengths(P, max_new_tokens, max_length)
    if max_len <= P:
        return input_ids
    if seed is None:  # torch.manual_seed governs the stream, like HF's torch.multinomial
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if do_sample else 0
    stop = _Stopper(max_len, eos_ids, stopping_criteria)
    if num_beams > 1:
        if do_sample:
            raise NotImplementedError("beam-sample is not one of the reference's decoding modes (eval scripts: num_beams=5, do_sample unset)")
        if streamer is not None:  # (transformers raises the same way)
            raise ValueError("`streamer` cannot be used with beam search. Make sure that `num_beams` is set to 1.")
        if attention_mask is not None and not bool(attention_mask.to(torch.bool).all()):
            raise NotImplementedError("beam search takes un-padded prompts (the reference's eval scripts decode one prompt at a time)")
        return _beam_search(model, input_ids, images, num_beams, max_len, eos_ids, pad, length_penalty, early_stopping, stop, fp8_weights)
    sel = dict(do_sample=bool(do_sample), temperature=float(temperature), top_k=int(top_k or 0), top_p=float(top_p), seed=seed)
    if streamer is not None:
        streamer.put(input_ids.cpu())
    if not use_cache:
        out = _sample_recompute(model, input_ids, images, attention_mask, max_len, eos_ids, pad, stop, sel, streamer)
    else:
        out = _sample_cached(model, input_ids, images, attention_mask, max_len, eos_ids, pad, stop, sel, use_graph, fp8_weights, streamer)
    if streamer is not None:
        streamer.end()
    return out


def _select(logits, V, sel, step):
    return O.select_tokens(logits, V, do_sample=sel["do_sample"], temperature=sel["temperature"], top_k=sel["top_k"], top_p=sel["top_p"],
                           seed=sel["seed"], step=step)


def _sample_recompute(model, ids, images, attention_mask, max_len, eos_ids, pad, stop, sel, streamer=None):
    """use_cache=False: full-sequence forward per token (the cross-check of the cached path)."""
    V = model.config.vocab_size
    unfinished = torch.ones(ids.shape[0], dtype=torch.bool, device=ids.device)
    step = 0
    while True:
        out = model.forward(input_ids=ids, attention_mask=attention_mask, images=images)
        logits = out.logits[:, -1, :].float().contiguous()
        nxt = _select(logits, V, sel, step).to(ids.device)
        if eos_ids:
            nxt = torch.where(unfinished, nxt, torch.full_like(nxt, pad))
        if streamer is not None:
            streamer.put(nxt.cpu())
        ids = torch.cat([ids, nxt[:, None]], dim=1)
        if attention_mask is not None:
            attention_mask = torch.cat([attention_mask, torch.ones_like(attention_mask[:, :1])], dim=1)
        unfinished = unfinished & ~stop(ids, None)
        step += 1
        if not bool(unfinished.any()):
            return ids


def _sample_cached(model, input_ids, images, attention_mask, max_len, eos_ids, pad, stop, sel, use_graph, fp8_weights, streamer=None):
    eng = model.engine
    B, P = input_ids.shape
    V = model.config.vocab_size
    max_new = max_len - P
    logits, cache = eng.prefill(input_ids, attention_mask, images, max_new)
    dev = logits.device
    graph = None
    if use_graph and logits.is_cuda and max_new > 2:
        graph, g_tok, g_logits = eng.capture_decode_graph(cache, fp8=fp8_weights)  # the decode step as one replayable HIP graph
    am = attention_mask.to(dev).to(torch.bool) if attention_mask is not None else None
    # right-padded prompts (ones then zeros; an extension - HF wants left padding): each row continues from its own length.
    # Anything else (HF's left padding, holes) follows transformers: new tokens are appended after the padded prompt, keys are the
    # valid positions only, rotary positions stay absolute (the engine's unpad / pad attention path + compacted KV cache).
    # ONE decision, the engine's (made on the device from the mask in forward(): a right-padded prefix keeps the lens fast path, anything
    # else - or engine.force_unpad - compacts the valid keys and sets cache.rpos): generate() follows it instead of classifying again
    padded = am is not None and not bool(am.all()) and cache.rpos is None
    ids = input_ids.to(dev)
    lens = am.sum(dim=1) if padded else None
    unfinished = torch.ones(B, dtype=torch.bool, device=dev)
    new = []
    for step in range(max_new):
        nxt = _select(logits, V, sel, step)
        if eos_ids:
            nxt = torch.where(unfinished, nxt, torch.full_like(nxt, pad))
        new.append(nxt)
        if streamer is not None:
            streamer.put(nxt.cpu())
        cur = torch.cat([ids, torch.stack(new, dim=1)], dim=1)  # (right-padded prompts: criteria see the pads in the middle)
        unfinished = unfinished & ~stop(cur, None)
        if not bool(unfinished.any()) or step + 1 == max_new:
            break
        if graph is not None:
            g_tok.copy_(nxt)
            graph.replay()
            logits = g_logits
        else:
            logits = eng.decode_step(nxt, cache, fp8=fp8_weights)
    if padded:  # right-padded prompts (an extension; HF wants left padding): each row's continuation starts at its own length
        cur = _compact(ids, lens, torch.stack(new, dim=1), pad)
    return cur.to(input_ids.device)


def _compact(ids, lens, new, pad):
    B, P = ids.shape
    out = torch.full((B, P + new.shape[1]), pad, dtype=ids.dtype, device=ids.device)
    for b in range(B):
        lb = int(lens[b])
        out[b, :lb] = ids[b, :lb]
        out[b, lb:lb + new.shape[1]] = new[b]
    return out


def _gather_beams(t, idx):
    """t [B, n, ...] gathered along dim 1 by idx [B, k] (HF `_gather_beams`)."""
    while idx.dim() < t.dim():
        idx = idx.unsqueeze(-1)
    return torch.gather(t, 1, idx.expand(-1, -1, *t.shape[2:]))


def _beam_search(model, input_ids, images, nb, max_len, eos_ids, pad, length_penalty, early_stopping, stop, fp8_weights):
    """Beam search as transformers' `_beam_search` runs it (do_sample=False, num_return_sequences=1).  The prompt is
    prefilled ONCE per prompt and its KV cache expanded to num_beams rows (the reference reaches the same state through
    `inputs_embeds.repeat_interleave(5)`, base_mmgpt.py:162-163, after computing one row); every step the cache rows are
    re-ordered by the surviving beams' parents (HF `_reorder_cache`) with one gather kernel per layer."""
    eng = model.engine
    B, P = input_ids.shape
    V = model.config.vocab_size
    max_new = max_len - P
    logits0, cache0 = eng.prefill(input_ids, None, images, max_new)
    dev = logits0.device
    ids = input_ids.to(dev)
    expand = torch.arange(B, device=dev).repeat_interleave(nb)
    cache = eng.expand_cache(cache0, expand)
    del cache0
    logits = torch.empty(B * nb, logits0.shape[1], dtype=torch.float32, device=dev)
    O.gather_rows2d(logits0, expand, logits)

    n_eos = len(eos_ids)
    keep = max(2, 1 + n_eos) * nb
    top_mask = torch.cat([torch.ones(nb, dtype=torch.bool), torch.zeros(keep - nb, dtype=torch.bool)]).to(dev)
    fill = (pad or eos_ids[0]) if eos_ids else -1  # HF `_beam_search`: `pad_token_id or eos_token_id[0] if ... else -1` (a pad id of 0 is falsy)
    running = torch.full((B, nb, max_len), fill, dtype=torch.int64, device=dev)
    running[:, :, :P] = ids[:, None, :]
    sequences = running.clone()
    running_scores = torch.zeros(B, nb, dtype=torch.float32, device=dev)
    running_scores[:, 1:] = NEG
    beam_scores = torch.full((B, nb), NEG, dtype=torch.float32, device=dev)
    seq_len = torch.zeros(B, nb, dtype=torch.int64, device=dev)      # generated length of every finished hypothesis
    finished = torch.zeros(B, nb, dtype=torch.bool, device=dev)
    unsat = torch.ones(B, 1, dtype=torch.bool, device=dev)          # "early-stop heuristic unsatisfied"
    batch_off = (torch.arange(B, device=dev) * nb)[:, None]
    cur = P
    while True:
        acc = O.log_softmax_rows(logits, V, row_bias=running_scores.reshape(-1).contiguous())  # log_probs + running_beam_scores
        top_lp, top_i = torch.topk(acc.view(B, nb * V), k=keep)
        parent = top_i // V
        tok = top_i % V
        cand = _gather_beams(running, parent)
        cand[:, :, cur] = tok
        hits = stop(cand[:, :, :cur + 1].reshape(B * keep, cur + 1), None).view(B, keep)
        # beams that continue: the best non-finished candidates
        run_lp = top_lp + hits.to(torch.float32) * NEG
        nxt_i = torch.topk(run_lp, k=nb)[1]
        running = _gather_beams(cand, nxt_i)
        running_scores = _gather_beams(run_lp, nxt_i)
        beam_idx = (_gather_beams(parent, nxt_i) + batch_off).reshape(-1)
        # finished hypotheses: only the top num_beams candidates may finish; keep the best num_beams overall
        just = hits & top_mask[None, :]
        fin_lp = top_lp / float((cur + 1 - P) ** length_penalty)
        full = finished.all(dim=-1, keepdim=True) & (early_stopping is True)
        fin_lp = fin_lp + full.to(torch.float32) * NEG + (~unsat).to(torch.float32) * NEG + (~just).to(torch.float32) * NEG
        m_seq = torch.cat([sequences, cand], dim=1)
        m_sc = torch.cat([beam_scores, fin_lp], dim=1)
        m_len = torch.cat([seq_len, torch.full((B, keep), cur + 1 - P, dtype=torch.int64, device=dev)], dim=1)
        m_fin = torch.cat([finished, just], dim=1)
        best = torch.topk(m_sc, k=nb)[1]
        sequences, beam_scores = _gather_beams(m_seq, best), _gather_beams(m_sc, best)
        seq_len, finished = _gather_beams(m_len, best), _gather_beams(m_fin, best)
        cur += 1
        # early-stop heuristic (early_stopping=False default: best attainable running score at the current length)
        hyp_len = (max_len - P) if (early_stopping == "never" and length_penalty > 0.0) else (cur - P)
        best_running = running_scores[:, :1] / float(hyp_len ** length_penalty)
        worst_fin = torch.where(finished, beam_scores.min(dim=1, keepdim=True)[0], torch.full_like(beam_scores, NEG))
        unsat = unsat & (best_running > worst_fin).any(dim=-1, keepdim=True)
        go_on = bool(unsat.any()) and not (bool(finished.all()) and early_stopping is True) and not bool(hits.all())
        if not go_on:
            break
        eng.reorder_cache(cache, beam_idx, cur - 1)
        logits = eng.decode_step(running[:, :, cur - 1].reshape(-1).contiguous(), cache, fp8=fp8_weights)
    out_len = P + int(seq_len[:, 0].max())
    return sequences[:, 0, :out_len].to(input_ids.device)
