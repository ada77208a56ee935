"""Data parallelism: plain replicas + bucketed gradient all-reduce over RCCL/xGMI, overlapped with the
backward.  The reference gets this from FSDP SHARD_GRAD_OP (pretrain.sh:32-33); with
288 GB HBM nothing needs sharding, so each rank keeps the full arena and only gradients move.

Buckets are the engine's natural units (one decoder / encoder layer = one contiguous range of the
gradient arena, ~405 MB bf16 for a Llama-7B layer): `engine.on_grads_ready(names)` fires as soon as a
layer's weight gradients are final; the bucket's all-reduce is enqueued on a dedicated communication
stream that waits on an event recorded on the compute stream, so it runs under the remaining backward
GEMMs.  xGMI is point-to-point (7 links/GPU): few, large collectives per step (35 for the 7B model)
rather than many small ones.  The mean over ranks (mean of per-rank mean losses, as the reference's DP
does) is folded into the optimizer's grad scale (1/world_size)."""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, engine, process_group=None, bucket_bytes_min=0):
        self.engine = engine
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.comm_stream = None
        self.pending = []
        self.n_collectives = 0
        self.bytes = 0
        engine.on_grads_ready = self._on_ready

    def _on_ready(self, names):
        if self.world == 1:
            return
        A = self.engine.arena
        if names is None:  # end of backward: join the communication stream
            self.finish()
            return
        names = [n for n in names if A.params[n].requires_grad]
        if not names:
            return
        off, num = A.range_of(names)
        buf = A.gflat[off: off + num]
        if buf.is_cuda:
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream(device=buf.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(buf.device))
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self.pending.append(w)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg)
        self.n_collectives += 1
        self.bytes += buf.numel() * buf.element_size()

    def finish(self):
        for w in self.pending:
            w.wait()  # makes the current (compute) stream wait for the collective
        self.pending = []

    @property
    def grad_scale(self):
        return 1.0 / self.world
