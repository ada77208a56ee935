"""Two data-parallel RANKS driving the REAL engine on one GPU (VERDICT r3 #7a).

The other multi-process tests use stand-ins for the engine (tests/test_dp_cpu.py, bench.py --dry-run) or a group of ONE rank
(test_geometry_gpu.py: real RCCL, but an all-reduce of one rank is the identity).  Here two processes share `cuda:0`; each builds
the model, runs HipEngine.forward / backward on ITS OWN batch with merlin_amd.dp.GradSync hooked in, so the engine's real per-bucket
"gradients final" callbacks fire and every bucket of the gradient arena is all-reduced ACROSS two ranks while the backward is still
running on the compute stream.  RCCL refuses two ranks on one device, so the process group is gloo on device tensors (same
torch.distributed calls, same side stream / event protocol in dp.py).  Checked: the reduced arena = the sum of the two ranks'
single-rank gradients (computed locally without GradSync), identical on both ranks; the collective sequence is identical although
rank 1's batch has NO image (its image-side buckets are zero-filled and still reported - a mismatch would deadlock, hence the
timeouts); gradient accumulation reduces once per window; one optimizer step with grad_scale = 1/world leaves both replicas equal."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        import torch.distributed as dist

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import sys

        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from merlin_amd.dp import GradSync
        from merlin_amd.optim import FusedAdamW
        from oracle import cases as C
        from test_model_gpu import _build, _to_dev

        dtype = torch.float16  # (gloo sums fp16 on every build; the arena all-reduce is dtype-agnostic)
        cfg, b_img = C.get_case("tiny_padbatch")
        _, b_txt = C.get_case("tiny_textonly")
        # rank 0: two samples with images; rank 1: text only (images = one zero image per sample is what the collator would send;
        # here no image tensors at all -> the engine's "untouched bucket" path)
        txt_only = dict(b_txt)
        keep = [i for i in range(b_txt["input_ids"].shape[0]) if int((b_txt["input_ids"][i] == cfg.im_patch_token).sum()) == 0]
        txt_only = dict(input_ids=b_txt["input_ids"][keep], attention_mask=b_txt["attention_mask"][keep], labels=b_txt["labels"][keep], images=None)
        batches = [b_img, txt_only]

        def dev(b):
            if b["images"] is None:
                return dict(input_ids=b["input_ids"].cuda(), attention_mask=b["attention_mask"].cuda(), labels=b["labels"].cuda(), images=None)
            return _to_dev(b)

        model = _build(cfg, dtype)
        eng = model.engine
        # ---- single-rank gradients of BOTH batches (no GradSync): the expectation ----
        local = []
        for b in batches:
            for p in model.parameters():
                p.grad = None
            model(**dev(b)).loss.backward()
            local.append(eng.arena.gflat.float().clone())
        want = (local[0] + local[1])
        # ---- the data-parallel step: this rank's batch, real bucket firing, all-reduce across the two ranks ----
        for p in model.parameters():
            p.grad = None
        sync = GradSync(eng)
        assert sync.active and sync.world == 2
        model(**dev(batches[rank])).loss.backward()
        torch.cuda.synchronize()
        got = eng.arena.gflat.float()
        scale = float(want.abs().max())
        err = float((got - want).abs().max()) / scale
        order = list(sync.order)
        n_coll = sync.n_collectives
        # every rank must hold the same reduced arena, bit for bit
        mine = eng.arena.gflat.clone()
        other = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(other, mine)
        same = bool(torch.equal(other[0], other[1]))
        orders = [None, None]
        dist.all_gather_object(orders, order)
        # ---- accumulation window of 2 micro-steps: one reduce, result = sum over ranks of (g(b_rank) * 2) ----
        for p in model.parameters():
            p.grad = None
        before = sync.n_collectives
        for i in range(2):
            with sync.accumulate(i, 2):
                model(**dev(batches[rank])).loss.backward()
        torch.cuda.synchronize()
        acc_err = float((eng.arena.gflat.float() - 2 * want).abs().max()) / scale
        acc_coll = sync.n_collectives - before
        # ---- an optimizer step with the mean folded into the grad scale: replicas stay identical ----
        FusedAdamW(eng, lr=1e-3).step(grad_scale=sync.grad_scale, max_grad_norm=1.0)
        torch.cuda.synchronize()
        w = eng.arena.flat.clone()
        ws = [torch.empty_like(w) for _ in range(world)]
        dist.all_gather(ws, w)
        q.put((rank, dict(err=err, same=same, n_coll=n_coll, order_equal=orders[0] == orders[1], n_buckets=len(order), acc_err=acc_err,
                          acc_coll=acc_coll, weights_equal=bool(torch.equal(ws[0], ws[1])),
                          nonzero=float(local[rank].abs().max()) > 0)))
        dist.destroy_process_group()
    except Exception as e:  # surface the failure instead of a queue timeout
        import traceback

        q.put((rank, dict(error=f"{e}\n{traceback.format_exc()}")))


def test_two_ranks_on_one_gpu_reduce_the_real_engines_buckets():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = dict(q.get(timeout=600) for _ in procs)
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    for r in (0, 1):
        assert "error" not in res[r], res[r]["error"]
    print(res)
    for r in (0, 1):
        d = res[r]
        assert d["nonzero"]
        assert d["err"] < 2e-3, d            # fp16 sum of two fp16 gradient arenas vs their fp32 sum
        assert d["same"] and d["order_equal"] and d["weights_equal"], d
        assert d["n_coll"] == d["n_buckets"] >= 7, d   # head, 2 decoder layers, embedding, projector, 2 live CLIP layers, CLIP embeddings
        assert d["acc_coll"] == d["n_buckets"] and d["acc_err"] < 4e-3, d
