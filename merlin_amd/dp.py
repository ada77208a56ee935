"""Data parallelism: plain replicas + bucketed gradient all-reduce over RCCL/xGMI, overlapped with the
backward.  The reference gets this from FSDP SHARD_GRAD_OP (pretrain.sh:32-33); with
288 GB HBM nothing needs sharding, so each rank keeps the full arena and only gradients move.

Buckets are the engine's natural units (one decoder / encoder layer = one contiguous range of the
gradient arena, ~405 MB bf16 for a Llama-7B layer): `engine.on_grads_ready(names)` fires as soon as a
layer's weight gradients are final; the bucket's all-reduce is enqueued on a dedicated communication
stream that waits on an event recorded on the compute stream, so it runs under the remaining backward
GEMMs.  xGMI is point-to-point (7 links/GPU): few, large collectives per step (35 + 23 CLIP layers for
the 7B model) rather than many small ones.  The mean over ranks (mean of per-rank mean losses, as the
reference's DP does) is folded into the optimizer's grad scale (1/world_size).

The engine fires EVERY trainable bucket on EVERY backward in one fixed order (buckets a rank's batch does
not touch - a text-only batch has no image gradients - are zero-filled and still reported), so all ranks
issue the same sequence of collectives whatever their data: the reference gets the same guarantee from
its `0 * projector(dummy_feature)` trick (base_mmgpt.py:109-113).

Gradient accumulation (pretrain.sh:18 trains with --gradient_accumulation_steps 8; SURVEY §8e "reduce only
on the k-th micro-step"): micro-steps run under `no_sync()` and only accumulate locally; the last one
all-reduces the accumulated sum.  A backward that would accumulate onto gradients that were already
all-reduced raises instead of silently counting the first micro-batch world^(k-1) times."""
from __future__ import annotations

import contextlib
import time

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, engine, process_group=None, force=False):
        """force=True runs the collectives even when world_size == 1 (exercises the stream / event path on one GPU)."""
        self.engine = engine
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.comm_stream = None
        self.pending = []
        self.n_collectives = 0
        self.bytes = 0
        self.order = []  # bucket (offset, numel) sequence of the last synced backward (tests compare it across ranks)
        self._sync = True
        self._reduced = False  # the gradient arena holds all-reduced sums (until the next fresh backward / optimizer step)
        self._reduced_at = None  # engine.weight_version when they were reduced
        self.timing = False      # bench.py: time every collective (events on the communication stream) and the exposed wait
        self._t_coll, self._t_wait, self._t_host = [], [], [0.0, 0.0]
        engine.on_grads_ready = self._on_ready
        engine.on_backward_begin = self._on_begin

    # ---- gradient accumulation ------------------------------------------------------------------
    @contextlib.contextmanager
    def no_sync(self):
        """Backwards inside this context accumulate locally; no collective is issued (DDP.no_sync semantics)."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def accumulate(self, micro_step: int, accumulation_steps: int):
        """Context for micro-step `micro_step` (0-based) of an accumulation window: syncs only on the last one."""
        if (micro_step + 1) % accumulation_steps == 0:
            return contextlib.nullcontext()
        return self.no_sync()

    # ---- engine callbacks -----------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = True):
        """Zero (or detach) the model's gradients and forget that they were reduced.  (`optimizer.zero_grad(set_to_none=False)` /
        `model.zero_grad(set_to_none=False)` leave `.grad` attached; the next backward then verifies on the device that the arena
        really holds zeros - this call saves that check.)"""
        for p in self.engine.arena.params.values():
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()
        self._reduced = False
        self._known_zero = True  # (settles the next backward's question without a pass over the arena)

    def _grads_all_zero(self) -> bool:
        """True when the whole gradient arena holds zeros.  EXACT (an OR over the bit patterns, mh_any_nonzero: a sum of squares flushes
        |g| < ~1e-23 - representable in bf16 - to zero and NaNs need care); one pass over the arena + a host read, only taken when nothing
        cheaper settles the question (_on_begin)."""
        g = self.engine.arena.gflat
        if g is None:
            return True
        if g.is_cuda and g.element_size() == 2:
            from . import ops as O

            flag = torch.zeros(1, dtype=torch.int32, device=g.device)
            O.any_nonzero(g, flag)
            return int(flag.item()) == 0
        return not bool(torch.count_nonzero(g.view(torch.int16) & 0x7FFF)) if g.element_size() == 2 else not bool(torch.count_nonzero(g))

    def _on_begin(self, fresh: bool):
        known_zero, self._known_zero = getattr(self, "_known_zero", False), False
        if fresh or known_zero:
            self._reduced = False
        elif self._reduced and self.active:
            # Attached gradients that were already all-reduced.  Accumulating onto them is only sound when they hold ZEROS
            # (optimizer.zero_grad(set_to_none=False) / model.zero_grad(set_to_none=False) after a step: `.grad` stays attached, the
            # arena cannot report `fresh`): that is checked on the device - a changed weight version alone is not enough, an
            # optimizer step WITHOUT zero_grad (or any in-place parameter edit) also bumps it and would otherwise let this backward
            # add local gradients onto world-summed ones and reduce them a second time (world * G_old + sum g_new).
            if self._grads_all_zero():
                self._reduced = False
            else:
                raise RuntimeError("merlin_amd.dp: this backward accumulates onto gradients that were already all-reduced; run all "
                                   "but the last micro-step of an accumulation window under GradSync.no_sync() (or zero_grad first)")
        if self._sync:
            self.order = []

    def _on_ready(self, names):
        if not self.active or not self._sync:
            return
        A = self.engine.arena
        if names is None:  # end of backward: join the communication stream
            self.finish()
            self._reduced = True
            self._reduced_at = self.engine.weight_version
            return
        names = [n for n in names if A.params[n].requires_grad]
        if not names:
            return
        off, num = A.range_of(names)
        buf = A.gflat[off: off + num]
        self.order.append((off, num))
        if buf.is_cuda:
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream(device=buf.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(buf.device))
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                if self.timing:
                    t0 = torch.cuda.Event(enable_timing=True)
                    t0.record(self.comm_stream)
                w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                if self.timing:
                    w.wait()  # (orders the communication stream - the current one here - behind the collective; no host wait)
                    t1 = torch.cuda.Event(enable_timing=True)
                    t1.record(self.comm_stream)
                    self._t_coll.append((t0, t1))
            self.pending.append(w)
        else:
            t0 = time.perf_counter()
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg)
            if self.timing:
                dt = (time.perf_counter() - t0) * 1e3
                self._t_host[0] += dt
                self._t_host[1] += dt  # synchronous on the host path: all of it is exposed
        self.n_collectives += 1
        self.bytes += buf.numel() * buf.element_size()

    def finish(self):
        timed = self.timing and self.pending and self.comm_stream is not None
        if timed:
            a = torch.cuda.Event(enable_timing=True)
            a.record(torch.cuda.current_stream())
        for w in self.pending:
            w.wait()  # makes the current (compute) stream wait for the collective
        if timed:
            b = torch.cuda.Event(enable_timing=True)
            b.record(torch.cuda.current_stream())
            self._t_wait.append((a, b))  # how long the compute stream stood still for the tail of the communication
        self.pending = []

    def pop_timing(self):
        """Call after a device synchronize: {comm_ms_total (sum over collectives, on the communication stream), comm_ms_exposed
        (time the compute stream waited at the end of backward), collectives, bytes} since the last call."""
        each = [a.elapsed_time(b) for a, b in self._t_coll]  # per collective, in issue order (steps x buckets)
        total = self._t_host[0] + sum(each)
        exposed = self._t_host[1] + sum(a.elapsed_time(b) for a, b in self._t_wait)
        out = dict(comm_ms_total=total, comm_ms_exposed=exposed, collectives=self.n_collectives, bytes=self.bytes, each_ms=each,
                   bucket_bytes=[n * self.engine.arena.gflat.element_size() for _, n in self.order])
        self._t_coll, self._t_wait, self._t_host = [], [], [0.0, 0.0]
        self.n_collectives = self.bytes = 0
        return out

    @property
    def grad_scale(self):
        return 1.0 / self.world
