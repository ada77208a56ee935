"""TEST INFRASTRUCTURE (oracle): CPU restatement of the decoding loops the reference's eval scripts reach through
transformers' GenerationMixin (eval_mmvet.py:101-120): greedy, multinomial sampling with the temperature / top-k / top-p
warpers, and beam search (transformers/generation/utils.py `_sample`, `_beam_search`, logits_process.py) - over ANY
`logits_fn(ids [B, L]) -> last-position logits [B, V]` (here: the CPU oracle's full forward, recomputed per token).

Pinning: greedy and beam search are checked against token ids produced by the REAL reference's `model.generate` in the build
container (oracle/make_gen_golden.py -> tests/golden/gen_*.json).  Sampling cannot be pinned to torch.multinomial's stream;
`warp_probs` (the warpers, sort-based exactly like HF) is pinned by comparing against transformers' own
TemperatureLogitsWarper/TopKLogitsWarper/TopPLogitsWarper in make_gen_golden.py, and `counter_uniform` restates the
counter-based uniform of mh_select_tokens so the device sampler can be replayed bit-for-bit in its choice of u.
Never imported by merlin_amd/."""
from __future__ import annotations

import numpy as np
import torch

NEG = -1.0e9
M64 = (1 << 64) - 1


def _splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)


def counter_uniform(seed: int, step: int, row: int) -> float:
    r = _splitmix64((seed & M64) ^ _splitmix64((step * 0x100000001B3 + row) & M64))
    return float(np.float32(r >> 40) * np.float32(1.0 / 16777216.0))


def warp_probs(logits: np.ndarray, temperature=1.0, top_k=0, top_p=1.0) -> np.ndarray:
    """HF's warper chain on one row -> probabilities (float64): TemperatureLogitsWarper, TopKLogitsWarper (remove
    scores < k-th largest), TopPLogitsWarper (sort ascending, remove cumulative_probs <= 1 - top_p, keep >= 1)."""
    z = logits.astype(np.float64) / float(temperature)
    if top_k and top_k < z.size:
        kth = np.sort(z)[-top_k]
        z = np.where(z < kth, -np.inf, z)
    if top_p < 1.0:
        order = np.argsort(z, kind="stable")  # ascending
        zs = z[order]
        p = np.exp(zs - zs.max())
        p /= p.sum()
        remove = np.cumsum(p) <= (1.0 - top_p)
        remove[-1:] = False
        z[order[remove]] = -np.inf
    p = np.exp(z - z.max())
    return p / p.sum()


def sample_from(probs: np.ndarray, u: float) -> int:
    """Inverse CDF in index order (the device sampler's rule): smallest i with cdf[i] > u * total."""
    c = np.cumsum(probs)
    i = int(np.searchsorted(c, u * c[-1], side="right"))
    return min(i, int(np.nonzero(probs > 0)[0][-1]))


def _stop(ids, max_length, eos_ids, user=()):
    done = torch.full((ids.shape[0],), ids.shape[1] >= max_length, dtype=torch.bool)
    for e in eos_ids:
        done |= ids[:, -1] == e
    for c in user:
        r = torch.as_tensor(c(ids, None))
        done |= r.bool() if r.dim() else r.bool().expand(ids.shape[0])
    return done


def sample(logits_fn, ids, max_new_tokens, eos_ids=(), pad=0, do_sample=False, temperature=1.0, top_k=50, top_p=1.0, seed=0, user=()):
    """transformers `_sample` (greedy when do_sample=False)."""
    max_length = ids.shape[1] + max_new_tokens
    unfinished = torch.ones(ids.shape[0], dtype=torch.bool)
    step = 0
    while True:
        logits = logits_fn(ids).float().numpy()
        if do_sample:
            nxt = [sample_from(warp_probs(logits[r], temperature, top_k, top_p), counter_uniform(seed, step, r)) for r in range(ids.shape[0])]
        else:
            nxt = logits.argmax(-1).tolist()
        nxt = torch.tensor(nxt, dtype=torch.int64)
        if eos_ids:
            nxt = torch.where(unfinished, nxt, torch.full_like(nxt, pad))
        ids = torch.cat([ids, nxt[:, None]], 1)
        unfinished &= ~_stop(ids, max_length, eos_ids, user)
        step += 1
        if not bool(unfinished.any()):
            return ids


def beam_search(logits_fn, ids, num_beams, max_new_tokens, eos_ids=(), pad=0, length_penalty=1.0, early_stopping=False, user=()):
    """transformers `_beam_search` (do_sample=False, num_return_sequences=1), one hypothesis list per prompt, plain Python
    bookkeeping (lists and floats) instead of the library's vectorised tensors - an independent formulation of the same rules:
    candidates = the (1 + n_eos) * num_beams best continuations by accumulated log-prob; candidates that hit a stopping
    criterion can finish only if they rank inside the first num_beams; finished score = sum_logprob / generated_len ** lp;
    the search stops when no running beam can beat the worst kept hypothesis (early_stopping=False heuristic: best running
    score / current generated length), when every candidate is stopped, or at max_length."""
    B, P = ids.shape
    max_length = P + max_new_tokens
    nb = num_beams
    keep = max(2, 1 + len(eos_ids)) * nb
    out = []
    for b in range(B):
        running = [(0.0 if i == 0 else NEG, ids[b].tolist()) for i in range(nb)]
        finished = []  # (score, tokens) best-first, at most nb; HF initialises nb placeholders at -1e9
        placeholders = nb
        unsat = True
        cur = P
        while True:
            logits = logits_fn(torch.tensor([seq for _, seq in running], dtype=torch.int64)).float()
            logp = torch.log_softmax(logits, -1).numpy().astype(np.float32)
            V = logp.shape[1]
            acc = (logp + np.array([s for s, _ in running], dtype=np.float32)[:, None]).reshape(-1)
            top = torch.topk(torch.from_numpy(acc), keep)
            cands = []
            for lp, i in zip(top.values.tolist(), top.indices.tolist()):
                seq = running[i // V][1] + [i % V]
                hit = bool(_stop(torch.tensor([seq]), max_length, eos_ids, user)[0])
                cands.append((lp, seq, hit))
            # running beams: best nb non-stopped candidates (stopped ones pushed to -1e9, order among equals = candidate order)
            run_sc = torch.tensor([lp + (NEG if hit else 0.0) for lp, _, hit in cands], dtype=torch.float32)
            nxt = torch.topk(run_sc, nb).indices.tolist()
            running = [(float(run_sc[j]), cands[j][1]) for j in nxt]
            # finished: candidates inside the first nb that were stopped
            full = (placeholders == 0) and early_stopping is True
            glen = cur + 1 - P
            fin_sc = []
            for rank, (lp, seq, hit) in enumerate(cands):
                sc = np.float32(lp) / np.float32(glen ** length_penalty)
                sc = float(sc) + (NEG if full else 0.0) + (NEG if not unsat else 0.0) + (NEG if not (hit and rank < nb) else 0.0)
                fin_sc.append(sc)
            merged = [(s, t, True) for s, t in finished] + [(NEG, None, False)] * placeholders + \
                     [(fin_sc[r], cands[r][1], cands[r][2] and r < nb) for r in range(keep)]
            order = torch.topk(torch.tensor([m[0] for m in merged], dtype=torch.float32), nb).indices.tolist()
            sel = [merged[j] for j in order]
            finished = [(s, t) for s, t, f in sel if f]
            placeholders = nb - len(finished)
            cur += 1
            best_running = np.float32(running[0][0]) / np.float32((cur - P) ** length_penalty)
            worst = min([s for s, _ in finished], default=NEG) if placeholders == 0 else NEG
            # HF: worst_finished_score = where(is_sent_finished, min(beam_scores), -1e9) per slot, any() over slots
            slots = [min([s for s, _ in finished] + [NEG] * placeholders)] * len(finished) + [NEG] * placeholders
            unsat = unsat and any(best_running > w for w in slots)
            del worst
            all_hit = all(h for _, _, h in cands)
            if not (unsat and not (placeholders == 0 and early_stopping is True) and not all_hit):
                break
        best = finished[0][1] if finished else running[0][1]
        out.append(best)
    L = max(len(t) for t in out)
    fill = (pad or eos_ids[0]) if eos_ids else -1  # HF: `pad_token_id or eos_token_id[0] if eos_token_id is not None else -1` (pad 0 is falsy)
    return torch.tensor([t + [fill] * (L - len(t)) for t in out], dtype=torch.int64)
