"""N>1 path on CPU: world_size-2 gloo run of the gradient bucketing / all-reduce logic (merlin_amd/dp.py)
over a real Arena layout, plus rank-dependent synthetic shards.  No GPU, no compute kernels."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeEngine:
    """Stands in for HipEngine.backward's bucket protocol on CPU tensors: on_backward_begin(fresh), then every bucket in
    the fixed order, then None."""

    def __init__(self, arena, buckets=None):
        self.arena = arena
        self.on_grads_ready = None
        self.on_backward_begin = None
        self.buckets = buckets
        self.fresh = True
        self.weight_version = 0  # HipEngine bumps it on every optimizer step / weight load

    def backward(self, value):
        """local gradient of this micro-step = `value` everywhere (written when fresh, accumulated otherwise)."""
        A = self.arena
        if self.on_backward_begin is not None:
            self.on_backward_begin(self.fresh)
        for names in self.buckets:
            for n in names:
                g = A.gview(n)
                if self.fresh:
                    g.fill_(value)
                else:
                    g.add_(value)
            self.on_grads_ready(names)
        self.on_grads_ready(None)
        self.fresh = False

    def zero_grad(self):
        self.fresh = True


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from merlin_amd.dp import GradSync
        from merlin_amd.model.arena import Arena

        names = [f"model.layers.{i}.{n}" for i in range(3) for n in ("a.weight", "b.weight", "norm.weight")] + ["lm_head.weight"]
        params = [(n, nn.Parameter(torch.zeros(130 if "norm" not in n else 7))) for n in names]
        params[4][1].requires_grad_(False)  # a frozen parameter inside layer 1
        A = Arena(params)
        A.flat = torch.zeros(A.total)
        A.gflat = torch.full((A.total,), float(rank + 1))
        eng = _FakeEngine(A)
        sync = GradSync(eng)
        assert sync.world == world and eng.on_grads_ready is not None
        # backward order: head first, then layers 2, 1 (layer 0 never reported -> must stay local)
        eng.on_grads_ready(["lm_head.weight"])
        eng.on_grads_ready([n for n in names if n.startswith("model.layers.2.")])
        eng.on_grads_ready([n for n in names if n.startswith("model.layers.1.")])
        eng.on_grads_ready(None)
        tot = float(sum(r + 1 for r in range(world)))
        ok = True
        for n in names:
            v = A.gview(n)
            want = tot if (n == "lm_head.weight" or n.startswith("model.layers.2.") or n.startswith("model.layers.1.")) else float(rank + 1)
            ok &= bool((v == want).all())
        ok &= sync.n_collectives == 3 and abs(sync.grad_scale - 1.0 / world) < 1e-12
        # per-rank synthetic shards differ (weak scaling: each rank draws its own batch)
        from merlin_amd import synth

        b = synth.interpair_batch(B=1, S=64, frames=1, base_vocab=100, P=4, image_size=14, rank=rank)
        ids = b["input_ids"].float().sum().reshape(1)
        gathered = [torch.zeros(1) for _ in range(world)]
        dist.all_gather(gathered, ids)
        ok &= len({float(x) for x in gathered}) == world
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def _run2(worker):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    return sorted(res)


def test_gradsync_world2_gloo():
    res = _run2(_worker)
    assert res == [(0, True), (1, True)], res


def _accum_worker(rank, world, port, q):
    """Gradient accumulation (pretrain.sh:18: --gradient_accumulation_steps 8; SURVEY §8e "reduce only on the k-th
    micro-step"): k micro-steps, only the last one all-reduces; result = sum over ranks of the sum over micro-steps."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from merlin_amd.dp import GradSync
        from merlin_amd.model.arena import Arena

        names = [f"model.layers.{i}.{n}" for i in range(2) for n in ("a.weight", "norm.weight")] + ["lm_head.weight"]
        A = Arena([(n, nn.Parameter(torch.zeros(130 if "norm" not in n else 7))) for n in names])
        A.flat = torch.zeros(A.total)
        A.gflat = torch.full((A.total,), 1234.0)  # stale values from "the previous step"
        buckets = [["lm_head.weight"], [n for n in names if ".1." in n], [n for n in names if ".0." in n]]
        eng = _FakeEngine(A, buckets)
        sync = GradSync(eng)
        ok = True
        k = 3
        for step in range(2):
            eng.zero_grad()
            before = sync.n_collectives
            for i in range(k):
                with sync.accumulate(i, k):
                    eng.backward(float((rank + 1) * (i + 1) + 10 * step))
                if i < k - 1:
                    ok &= sync.n_collectives == before  # nothing reduced on the first k-1 micro-steps
            want = sum(sum((r + 1) * (i + 1) + 10 * step for i in range(k)) for r in range(world))
            ok &= sync.n_collectives == before + len(buckets)
            ok &= all(bool((A.gview(n) == float(want)).all()) for n in names)
            ok &= sync.order == [A.range_of(b) for b in buckets]  # the fixed bucket order (identical on every rank)
        # accumulating onto all-reduced gradients without no_sync is an error, not silent double counting
        try:
            eng.backward(1.0)
            ok = False
        except RuntimeError as e:
            ok &= "no_sync" in str(e)
        # an optimizer step WITHOUT zero_grad (or any in-place parameter edit) bumps the weight version while the arena still
        # holds the all-reduced sums: accumulating onto them must still raise (it used to be let through: world * G_old + sum g_new)
        eng.weight_version += 1
        try:
            eng.backward(1.0)
            ok = False
        except RuntimeError as e:
            ok &= "no_sync" in str(e)
        # after an optimizer step + zero_grad(set_to_none=False) the gradients are zero but still ATTACHED (the arena
        # reports fresh = False); that is a valid new window, recognised by the arena holding zeros ...
        A.gflat.zero_()
        eng.backward(float(rank + 1))
        ok &= all(bool((A.gview(n) == float(sum(r + 1 for r in range(world)))).all()) for n in names)
        # ... and without a step (skipped update) through GradSync.zero_grad()
        sync.zero_grad(set_to_none=False)
        A.gflat.zero_()  # (this CPU stand-in has no .grad views attached for zero_grad to reach)
        eng.backward(2.0)
        ok &= all(bool((A.gview(n) == 2.0 * world).all()) for n in names)
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_gradsync_accumulation_reduces_only_on_last_micro_step():
    res = _run2(_accum_worker)
    assert res == [(0, True), (1, True)], res


def test_arena_layout_and_fused_spans():
    from merlin_amd.model.arena import ALIGN, Arena

    ps = [("q", nn.Parameter(torch.zeros(4, 64))), ("k", nn.Parameter(torch.zeros(4, 64))), ("v", nn.Parameter(torch.zeros(4, 64))),
          ("n", nn.Parameter(torch.zeros(5))), ("head", nn.Parameter(torch.zeros(3, 64)))]
    A = Arena(ps, alloc_numel={"head": 4 * 64})
    assert A.offset["k"] == 256 and A.offset["n"] == 768 and A.offset["head"] == 768 + ALIGN and A.total == 768 + ALIGN + 256
    A.flat = torch.arange(A.total, dtype=torch.float32)
    assert A.span("q", "v", (12, 64)).shape == (12, 64) and float(A.span("q", "v", (12, 64))[4, 0]) == 256.0
    assert A.view("head", numel=256, shape=(4, 64)).shape == (4, 64)
    assert A.range_of(["q", "k", "v"]) == (0, 768)
    with pytest.raises(AssertionError):
        A.span("v", "head", (1, 1))
