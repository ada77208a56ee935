#!/usr/bin/env python
"""bench.py - the hot path's headline benchmark (BASELINE.json metric) on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3|cfg2] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
`python bench.py --gpus N` with N > 1 and no torchrun environment re-launches itself as N workers (one per GPU, rendezvous on
127.0.0.1) and still prints ONE line; under torchrun it uses the environment it is given.  `--dry-run` swaps the model for a CPU
stand-in over gloo (tests/test_bench_cpu.py: launcher, GradSync, timing protocol and the JSON line at world size 2, no GPU).

A "step" = one full training step of the hot path on one synthetic interpair batch per GPU
(BASELINE cfg 3/4: B=8 sequences x S=4096 = 6 x 336-px frames + trajectory text, ViT-L/14 + mlp
projector + Llama-7B, random-init weights from the build's generator, bf16):
    forward (ViT -> projector -> splice -> 32 decoder layers -> lm_head -> shifted CE)
    + backward (all weight gradients; activations stay resident in the 288 GB of HBM by default, --recompute switches to
      the reference's per-layer gradient checkpointing)
    + [N>1] bucketed RCCL all-reduce of the 14 GB gradient arena, overlapped with the backward
    + global-norm clipping + fused AdamW (LLRD groups, cosine schedule) over the parameter arena.
Inputs are resident in HBM before the timed region.  value = N * B * S / (max-over-ranks step time).
After the timed training steps the same batch is also timed FORWARD-ONLY (`forward_only` in the JSON line: the north star's
">= 40 % of the bf16 MFMA roofline on the fused ViT+LLM forward" is quoted on that leg).
One JSON line is printed by rank 0, with `roofline` (dominant kernel = the MFMA GEMM, timed per launch with
HIP events on the launch stream inside the timed steps) and `cpu_baseline` (the CPU oracle = a port of the
reference's CPU forward, timed on this box's host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# dmabuf IPC only on these hosts: without it RCCL's peer mappings fail with `hipIpcGetMemHandle: invalid argument`.  Exported on the GPU boxes already; set here as
# well (before the HIP runtime comes up) so that a bare torchrun environment cannot lose it.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md), not the 2:1-sparse figure
PEAK_FP8_TFLOPS = 5000.0   # dense fp8 MFMA peak

LLAMA_7B = dict(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                rms_norm_eps=1e-6, rope_theta=10000.0, max_position_embeddings=8192)
VIT_L_336 = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336, patch_size=14)


def algorithmic_flops_fwd(B, S, n_img, vit_layers_used=23):
    """2*MAC, causal attention at half of S^2, dead ViT layer and recompute not counted."""
    d, ff, L, V = 4096, 11008, 32, 32003
    llama = B * (L * (S * (8 * d * d + 6 * d * ff) + 2 * S * S * d) + 2 * S * d * V)
    vd, vff, Sv = 1024, 4096, 577
    per_layer = Sv * (8 * vd * vd + 4 * vd * vff) + 4 * Sv * Sv * vd
    vit = n_img * (2 * 576 * 588 * vd + vit_layers_used * per_layer)
    proj = n_img * 2 * 576 * vd * d
    return llama + vit + proj


def _full_depth_params(P1, cfg, own_memory):
    """Timing only: a FULL-DEPTH parameter set from ONE layer's random draw per tower, without drawing 7 B normals on the host first.
    own_memory=True: every layer gets its OWN tensors (layer 0's draw scaled by 1 + i/1024 - a multi-threaded copy, 28 GB of fp32 in all), so no
    weight byte is ever read twice by one forward; False: layer i points at layer 0's tensors (one decoder layer is 0.8 GB in fp32 - far larger
    than the host caches, so that flatters nothing either, but it is stated in the line)."""
    out = dict(P1)
    for k in list(P1):
        for pre, n in (("model.layers.", cfg.num_hidden_layers), ("model.vision_tower.vision_tower.vision_model.encoder.layers.", cfg.v_num_hidden_layers)):
            if k.startswith(pre + "0."):
                for i in range(1, n):
                    out[pre + f"{i}." + k[len(pre) + 2:]] = (P1[k] * (1.0 + i / 1024.0)) if (own_memory and P1[k].dim() > 1) else P1[k]
    return out


def _pick_threads(rows):
    """torch's default pool on these hosts (`threads_default`, all hardware threads the container sees) is not the fastest for the oracle's
    matmuls: a decoder-MLP-sized product with `rows` rows is timed at a few pool sizes and the fastest is used (reported as `cores`)."""
    all_threads = torch.get_num_threads()
    xa, wa = torch.randn(rows, 4096), torch.randn(11008, 4096)
    best = (None, float("inf"))
    for nt in sorted({all_threads, max(1, all_threads // 2), max(1, all_threads // 4), max(1, all_threads // 8), min(all_threads, 16)}, reverse=True):
        torch.set_num_threads(nt)
        torch.nn.functional.linear(xa, wa)
        t0 = time.perf_counter()
        for _ in range(3):
            torch.nn.functional.linear(xa, wa)
        dt_ = time.perf_counter() - t0
        if dt_ < best[1]:
            best = (nt, dt_)
    torch.set_num_threads(best[0])
    return best[0]


def cpu_baseline(seconds_budget=40.0, s4096=True):
    """The CPU oracle (oracle/ref_cpu.py = a port of the reference's CPU forward, pinned to the reference by golden vectors) timed on this
    box's host cores, fp32 (SURVEY 8d "CPU reference timing"):
      * `value` = ONE sequence of the metric's own configuration (BASELINE cfg 3: S = 4096, 6 x 336-px frames; 24-layer ViT-L with 23 live
        layers + projector + splice + 32-layer Llama-7B + lm_head + shifted CE), full depth, MEASURED: one timed forward (about 100 s of CPU
        work - the bounded sample is one of the step's eight sequences; the forward has no cross-sequence term, so tokens/s is unaffected);
      * `cfg1` = BASELINE cfg 1 at full depth (one image + 32-token caption, S = 613): 1 warm-up + 3 timed forwards, median."""
    import statistics

    from merlin_amd import synth
    from oracle import ref_cpu as R

    t_begin = time.perf_counter()
    torch.manual_seed(0)
    all_threads = torch.get_num_threads()
    try:
        import psutil

        ram_gb = psutil.virtual_memory().available / 1e9
    except Exception:
        ram_gb = 0.0
    own = ram_gb > 48.0  # 28.2 GB of fp32 weights + the S = 4096 activations (the [1, 32, S, S] scores are 2.1 GB a copy)
    one = R.OracleConfig(num_hidden_layers=1, v_num_hidden_layers=1)
    full = R.OracleConfig()
    P1 = {k: torch.empty(s).normal_(0, 0.02) if len(s) > 1 else torch.ones(s) for k, s in R.param_shapes(one).items()}
    P = _full_depth_params(P1, full, own)
    weights_note = ("every layer has its own tensors (one random draw per tower, scaled per layer: 28 GB of fp32, no byte read twice)" if own else
                    f"one layer's random weights aliased across the depth (host RAM available {ram_gb:.0f} GB < 48 GB) - timing only")

    def timed(batch):
        t0 = time.perf_counter()
        with torch.no_grad():
            R.forward(P, full, batch["input_ids"], batch["attention_mask"], batch["labels"], batch["images"])
        return time.perf_counter() - t0

    # ---- cfg 1, full depth ----
    nt1 = _pick_threads(613)
    b1 = synth.single_image_batch()
    S1 = int(b1["input_ids"].shape[1])
    first = timed(b1)
    n_timed = 3 if first * 4 <= seconds_budget else 1
    times = [timed(b1) for _ in range(n_timed)]
    med = statistics.median(times)
    cfg1 = {"value": round(S1 / med, 3), "unit": "tokens/s", "cores": nt1, "seconds_per_forward": round(med, 3), "warmup_forwards": 1, "timed_forwards": n_timed,
            "sample": f"BASELINE cfg 1 at full depth (1 x 336px image + 32-token caption, S={S1}), fp32 forward, median of {n_timed} (min {min(times):.2f} s, max {max(times):.2f} s)"}
    out = {"unit": "tokens/s", "kind": "port", "threads_default": all_threads, "weights": weights_note, "cfg1": cfg1,
           "cores_note": "cores = torch intra-op threads used, the fastest of {T, T/2, T/4, T/8, 16} (T = threads_default, the pool torch sizes to every "
                         "hardware thread the container sees) on a decoder-MLP-sized matmul of the leg's row count, timed just before the leg"}
    if not s4096:
        out.update(value=cfg1["value"], cores=nt1, sample=cfg1["sample"], wall_s=round(time.perf_counter() - t_begin, 1))
        return out
    # ---- the metric's configuration: one cfg-3 sequence, full depth, measured ----
    nt3 = _pick_threads(4096)
    b3 = synth.interpair_batch(B=1, S=4096)
    t3 = timed(b3)
    out.update(value=round(4096 / t3, 3), cores=nt3, seconds_per_forward=round(t3, 2), warmup_forwards=0, timed_forwards=1, measured=True,
               sample="1 of the step's 8 interpair sequences (BASELINE cfg 3: S=4096, 6 x 336px frames; ViT-L 23 live layers + mlp projector + splice + 32 decoder "
                      f"layers + lm_head + shifted CE), fp32 forward at FULL depth, one timed forward = {t3:.1f} s on {nt3} threads (pool and pages warmed "
                      "by the cfg-1 forwards just before)",
               wall_s=round(time.perf_counter() - t_begin, 1))
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg3", choices=["cfg3", "cfg2", "cfg3-ragged", "cfg5", "cfg5-bf16"],
                    help="cfg3 = BASELINE metric config (default); cfg3-ragged = same with ragged lengths + key padding; "
                         "cfg5 = BASELINE configs[4]: S=8192 interleave (4 images + long text), decoder GEMMs (forward, dgrad, wgrad) on the "
                         "fp8 MFMA weight path; cfg5-bf16 = the same shape with bf16 weights")
    ap.add_argument("--fp8-train", action="store_true", help="run the chosen config with the fp8 training step (cfg5 implies it)")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--recompute", action="store_true", help="recompute each layer's forward in backward (the reference's "
                    "gradient checkpointing) instead of keeping activations resident in the 288 GB of HBM")
    ap.add_argument("--fwd-only", action="store_true")
    ap.add_argument("--no-forward-leg", action="store_true", help="skip the extra forward-only timing after the training steps")
    ap.add_argument("--no-extras", action="store_true", help="skip the cfg-2 (S=613) row of the N=1 line's `extras`")
    ap.add_argument("--fp8-forward", action="store_true", help="with --fwd-only: decoder Linears on the scaled-fp8 MFMA (e4m3 operands, "
                    "per-row scales); NOT the headline configuration (dtype field says so)")
    ap.add_argument("--fp32-residual", action="store_true", help="(default since round 4, kept for old command lines) both towers' residual "
                    "streams in fp32 (engine.fp32_residual): the configuration whose parity is quoted")
    ap.add_argument("--residual-16bit", action="store_true", help="16-bit residual streams (engine.fp32_residual = False: the reference's own "
                    "bf16 training numerics, -1.4 % step time); NOT the headline configuration")
    ap.add_argument("--dry-run", action="store_true", help="CPU stand-in for the model over gloo: exercises the launcher, GradSync, the "
                    "barrier / max-over-ranks timing and the JSON line without a GPU (tests/test_bench_cpu.py); the numbers mean nothing")
    ap.add_argument("--force-dp", action="store_true", help="N=1 only: run the N>1 code path (RCCL process group of one rank, GradSync on its "
                    "communication stream, per-rank gathers, headroom check) on the one GPU - a hardware check of the multi-GPU plumbing")
    ap.add_argument("--min-free-gb", type=float, default=10.0, help="N>1: if less HBM than this stays free next to RCCL's buffers after the "
                    "first warm-up step, every rank lowers its resident-activation footprint: engine.mem_level 1 (normed GEMM operands re-derived in "
                    "backward, -20 GB), 2 (+ SwiGLU outputs of 16 layers recomputed, -31 GB), then full layer recompute (`recompute_fallback`)")
    ap.add_argument("--mem-level", type=int, default=0, choices=[0, 1, 2], help="engine.mem_level to start from (0 = keep every activation resident)")
    ap.add_argument("--emulate-comm", type=int, default=0, metavar="BLOCKS", help="N=1 only, measurement aid for the N>1 case no node is available for: "
                    "as every gradient bucket becomes final, a side stream streams it twice through BLOCKS workgroups (tools/probes/hbm_probe.so: the "
                    "CU and HBM footprint of a ring all-reduce's kernels - RCCL itself moves nothing in a group of one rank); the line reports the "
                    "bytes and the time the compute stream waited at the end of backward")
    ap.add_argument("--rotate", type=int, default=4, metavar="N", help="distinct seeded batches cycled through warm-up and the timed steps (default 4: every "
                    "step sees other token ids and images than the step before, as in training; all of them resident in HBM before the timed region)")
    ap.add_argument("--same-batch", action="store_true", help="feed ONE batch to every step (rounds 1-5; the loss then falls to ~0.3 inside the run and "
                    "dlogits / gradients shrink: A/B against the rotated default in profiles/r06_batch_rotation_ab.txt)")
    ap.add_argument("--no-cfg5-extra", action="store_true", help="skip the cfg-5 (S=8192 interleave, fp8 weight path) row of the N=1 line's `extras`")
    return ap.parse_args(argv)


def self_launch(args, argv):
    """`python bench.py --gpus N` outside a torchrun environment: re-launch as N workers of ONE node (one process per GPU,
    rendezvous on 127.0.0.1, RCCL over xGMI), the same command line the driver uses; rank 0 of the children prints the line."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, args.gpus))))
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


class _DryEngine:
    """--dry-run: the bucket protocol of HipEngine.backward on CPU tensors (a few small matmuls as 'compute'), so that the REAL
    GradSync, launcher, timing protocol and JSON assembly run at world size > 1 without a GPU."""

    def __init__(self, rank):
        from merlin_amd.model.arena import Arena

        names = [f"model.layers.{i}.{n}" for i in range(4) for n in ("a.weight", "b.weight")] + ["lm_head.weight"]
        self.arena = Arena([(n, torch.nn.Parameter(torch.zeros(64, 64))) for n in names])
        self.arena.flat = torch.zeros(self.arena.total)
        self.arena.gflat = torch.zeros(self.arena.total)
        self.buckets = [["lm_head.weight"]] + [[n for n in names if f".{i}." in n] for i in reversed(range(4))]
        self.on_grads_ready = self.on_backward_begin = None
        self.weight_version = 0
        self.save_activations = True
        self.x = torch.randn(64, 64, generator=torch.Generator().manual_seed(rank))

    def step(self, sync):
        if self.on_backward_begin is not None:
            self.on_backward_begin(True)
        for names in self.buckets:
            for n in names:
                self.arena.gview(n).copy_(self.x @ self.x.t())
            if self.on_grads_ready is not None:
                self.on_grads_ready(names)
        if self.on_grads_ready is not None:
            self.on_grads_ready(None)
        self.arena.flat.add_(self.arena.gflat, alpha=-1e-3 * (sync.grad_scale if sync else 1.0))
        self.weight_version += 1
        return self.arena.gflat.sum()


def kernel_source_stamp():
    """sha256 over the GEMM kernel sources (the kernels `roofline.traffic` is about): profiles/*traffic*.json files carry it, and a PMC
    summary taken on other GEMM kernels is not quoted."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "merlin_amd", "csrc")
    for f in ("gemm.hip", "gemm256.hip", "gemm_w4.hip", "gemm_common.h", "mh_common.h"):
        with open(os.path.join(d, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def traffic_summary_path():
    """The newest per-round PMC summary, profiles/rNN_gemm_traffic.json (tools/pmc_step_traffic.sh writes it)."""
    import glob

    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_gemm_traffic.json")))
    return found[-1] if found else None


def read_traffic(args):
    """HBM/fabric traffic of the dominant kernel comes from separate rocprofv3 --pmc passes of this same command
    (tools/pmc_step_traffic.sh; PMC collection cannot run inside the timed process).  The committed summary is quoted only when
    it was taken on the kernel sources of THIS build (stamp) and for the default configuration."""
    if args.config != "cfg3" or args.fwd_only or args.fp8_train or args.recompute or args.dry_run or args.residual_16bit:
        return None, None
    path = traffic_summary_path()
    name = "profiles/" + os.path.basename(path) if path else "profiles/r*_gemm_traffic.json"
    try:
        with open(path) as f:
            t = json.load(f)
    except Exception:
        return None, f"no PMC summary ({name})"
    if t.get("kernel_source_stamp") != kernel_source_stamp():
        return None, f"{name} was taken on other kernel sources (stamp {t.get('kernel_source_stamp')}): not quoted"
    return round(t["traffic_bytes_per_launch"] / 1e9, 3), f"GB per GEMM kernel launch (L2<->fabric incl. Infinity-Cache hits, PMC: {name}: " \
        f"{t.get('traffic_bytes_per_step', 0) / 1e12:.2f} TB per step = {t.get('traffic_over_algorithmic', 0):.2f} x the algorithmic " \
        f"{t.get('algorithmic_bytes_per_step', 0) / 1e12:.2f} TB; a bench `launch` below is one GEMM call = 1-3 kernel launches)"


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, argv)
    if os.environ.get("MH_GEMM_FORCE") and not args.dry_run:  # A/B arm selection for kernel development (see mh_gemm_force_kernel)
        from merlin_amd import ops as _O
        _O.gemm_force_kernel(int(os.environ["MH_GEMM_FORCE"]))

    # The result line must be the only thing on stdout: RCCL prints a version banner there (C stdio, flushed at exit, after our line) and
    # so might any other library.  Keep the real stdout aside for the one JSON line and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    dry = args.dry_run
    if dry:
        dev = torch.device("cpu")
        dev_sync = lambda: None  # noqa: E731
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        dev_sync = torch.cuda.synchronize
    dp = world > 1 or args.force_dp  # data-parallel plumbing active
    if dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        rccl_log = None
        if dry:
            dist.init_process_group("gloo")
        else:
            if rank == 0 and "NCCL_DEBUG" not in os.environ:
                # RCCL's own choice of algorithm / protocol per collective size: its TUNING lines ("<coll>: N Bytes -> Algo a proto p time t"), rank 0
                # only, into a file (one line per enqueued collective: ~35 per step, microseconds each), parsed once after the timed region
                rccl_log = f"/tmp/mh_rccl_tuning_{os.getpid()}.log"
                os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="TUNING", NCCL_DEBUG_FILE=rccl_log)
            dist.init_process_group("nccl", device_id=dev)

    def all_max(x):
        if not dp:
            return float(x)
        t = torch.tensor([float(x)], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t)

    def all_min(x):
        return -all_max(-float(x))

    def all_sum_int(x):
        if not dp:
            return int(x)
        t = torch.tensor([int(x)], device=dev, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t)

    from merlin_amd import synth
    from merlin_amd.dp import GradSync

    O = None
    model = None
    emu = None
    recompute_fallback = False
    hbm_info = {}
    if dry:
        eng = _DryEngine(rank)
        B, S, n_img, n_tok = 2, 64, 0, 128
        workload = "DRY RUN: CPU stand-in over gloo (launcher / GradSync / timing protocol only)"
        sync = GradSync(eng, force=args.force_dp) if dp else None

        def step():
            return eng.step(sync)
    else:
        from merlin_amd import ops as O
        from merlin_amd.model.llama_mmgpt import build_synthetic_model
        from merlin_amd.optim import FusedAdamW, cosine_with_warmup, vit_lr_scale

        assert O.arch_ok(local_rank), "bench.py needs a gfx950 (MI355X) device"
        model = build_synthetic_model(LLAMA_7B, VIT_L_336, projector="mlp", dtype=torch.bfloat16, device=dev, seed=0)
        eng = model.engine
        eng.save_activations = not args.recompute
        eng.mem_level = args.mem_level
        eng.fp32_residual = not args.residual_16bit
        if args.fp8_forward:
            assert args.fwd_only, "--fp8-forward is forward-only"
            model.fp8_forward = True
        n_rot = 1 if (args.same_batch or args.fwd_only) else max(1, args.rotate)
        more = []  # batches 1 .. n_rot-1 of the rotation (batch j of rank r is drawn with the seeds of "rank" r + world * j: no two alike in the job)
        if args.config == "cfg3":
            B = args.batch or 8
            batch = synth.interpair_batch(B=B, S=4096, rank=rank)
            more = [synth.interpair_batch(B=B, S=4096, rank=rank + world * j) for j in range(1, n_rot)]
            workload = f"interpair: B={B}/GPU x S=4096 (6 x 336px frames + trajectory text), ViT-L/14-336 + mlp projector + Llama-7B"
        elif args.config == "cfg3-ragged":
            B = args.batch or 8
            batch = synth.interpair_batch(B=B, S=4096, rank=rank, ragged=True)
            more = [synth.interpair_batch(B=B, S=4096, rank=rank + world * j, ragged=True) for j in range(1, n_rot)]
            workload = f"interpair ragged: B={B}/GPU, lengths <= 4096 right-padded (key-padding branch), 6 frames, ViT-L/14-336 + mlp + Llama-7B"
        elif args.config in ("cfg5", "cfg5-bf16"):
            B = args.batch or 4
            batch = synth.interleave_batch(B=B, S=8192, n_images=4, rank=rank)
            more = [synth.interleave_batch(B=B, S=8192, n_images=4, rank=rank + world * j) for j in range(1, n_rot)]
            if args.config == "cfg5":
                args.fp8_train = True
                workload = f"interleave (MMC4-style): B={B}/GPU x S=8192, 4 images per document, fp8 (e4m3) MFMA weight path for the decoder's Linear layers and lm_head"
            else:
                workload = f"interleave (MMC4-style): B={B}/GPU x S=8192, 4 images per document, bf16 weights (NOT cfg 5's fp8 weight path)"
        else:
            B = 1
            batch = synth.single_image_batch()
            workload = "single 336px image + 32-token caption (S=613), ViT-L/14-336 + mlp projector + Llama-7B"
        if args.fp8_train:
            model.fp8_training = True
            # cfg 5 as SURVEY §8d states it: fp8 for ALL Llama / ViT Linear weights.  The tower leg costs +0.6 % (its K = 1024 products gain nothing
            # from fp8 and pay the quantisation passes: profiles/r03_fp8_parts_ab.txt), so the engine's own default leaves it off; the benchmark
            # of the configuration switches it on.
            model.engine.fp8_tower = True
            if os.environ.get("MH_FP8_PARTS"):  # A/B: "decoder" = tower and head stay 16-bit; "decoder,tower" / "decoder,head"
                parts = os.environ["MH_FP8_PARTS"].split(",")
                model.engine.fp8_tower, model.engine.fp8_head = "tower" in parts, "head" in parts
        S = batch["input_ids"].shape[1]
        n_img = sum(int(im.shape[0]) for im in batch["images"])
        n_tok = int(batch["attention_mask"].sum())  # the metric counts sequence positions, padding excluded

        def to_dev(b):
            return dict(input_ids=b["input_ids"].to(dev), attention_mask=b["attention_mask"].to(dev), labels=b["labels"].to(dev),
                        images=[im.to(dev) for im in b["images"]])

        dbatch = to_dev(batch)
        dbatches = [dbatch] + [to_dev(b) for b in more]  # (cfg 2 is one fixed sample: nothing to rotate)
        n_tok_rot = [n_tok] + [int(b["attention_mask"].sum()) for b in more]
        del more
        # the reference's recipe (pretrain.sh:23-29): --llrd, lr 5e-5, beta2 0.95, wd 0.05, cosine warm-up; HF's default
        # max_grad_norm=1.0 clipping - all of it inside the timed step
        opt = FusedAdamW(eng, lr=5e-5, betas=(0.9, 0.95), weight_decay=0.05, lr_scale_fn=vit_lr_scale)
        it = [0]
        sync = GradSync(eng, force=args.force_dp) if dp else None

        emu = None
        if args.emulate_comm and not dp:
            import ctypes

            plib = ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "hbm_probe.so"))
            emu = {"stream": torch.cuda.Stream(device=dev), "sink": torch.zeros(1 << 16, dtype=torch.int32, device=dev), "bytes": 0, "wait": [], "n": 0}

            def _emu_ready(names):
                A = eng.arena
                cur = torch.cuda.current_stream(dev)
                if names is None:  # end of backward: the optimizer must see "reduced" gradients
                    a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a_.record(cur)
                    cur.wait_stream(emu["stream"])
                    b_.record(cur)
                    emu["wait"].append((a_, b_))
                    return
                names = [n for n in names if A.params[n].requires_grad]
                if not names:
                    return
                off, num = A.range_of(names)
                buf = A.gflat[off: off + num]
                ev = torch.cuda.Event()
                ev.record(cur)
                emu["stream"].wait_event(ev)
                for _ in range(2):
                    plib.hbm_read(ctypes.c_void_p(buf.data_ptr()), ctypes.c_int64(buf.numel() * buf.element_size()), ctypes.c_void_p(emu["sink"].data_ptr()),
                                  args.emulate_comm, 4, ctypes.c_void_p(emu["stream"].cuda_stream))
                emu["bytes"] += 2 * buf.numel() * buf.element_size()
                emu["n"] += 1

            eng.on_grads_ready = _emu_ready

        def train_step(db):
            out = model(**db)
            out.loss.backward()
            # (GradSync joins the communication stream at the end of backward: the clip sees the all-reduced gradients)
            opt.step(grad_scale=(sync.grad_scale if sync else 1.0), max_grad_norm=1.0, lr_mult=cosine_with_warmup(it[0] + 10, 1000, 0.01))
            it[0] += 1
            opt.zero_grad()
            return out.loss

        calls = [0]

        def step():
            if args.fwd_only:
                with torch.no_grad():
                    return model(**dbatch).loss
            db = dbatches[calls[0] % len(dbatches)]
            calls[0] += 1
            return train_step(db)

    loss_first = None
    for wi in range(args.warmup):
        tw = time.perf_counter()
        loss = step()
        if wi == 0:
            loss_first = loss.detach().clone()  # read back after the timed region
        if os.environ.get("MH_BENCH_PER_STEP"):  # warm-up profile (stderr): how many steps until the step time is flat
            dev_sync()
            print(f"[bench] warm-up step {wi}: {(time.perf_counter() - tw) * 1e3:.1f} ms", file=sys.stderr, flush=True)
        if wi == 0 and dp and not dry and not args.fwd_only and eng.save_activations:
            # RCCL has allocated its channels / staging buffers by now (the first all-reduces ran): with activations resident the
            # step peaks at ~255 of 288 GB on one GPU.  If less than --min-free-gb stays free on ANY rank, all ranks step down together: first
            # engine.mem_level 1 / 2 (activations the backward re-derives: a slope of ~0.5 % / ~1 % step time), full layer recompute last.
            dev_sync()
            free_b, total_b = torch.cuda.mem_get_info(dev)
            headroom = (free_b + torch.cuda.memory_reserved(dev) - torch.cuda.max_memory_allocated(dev)) / 1e9
            hbm_info = {"hbm_total_gb": round(total_b / 1e9, 1), "hbm_headroom_gb_after_first_step": round(all_min(headroom), 1)}
            short = args.min_free_gb - hbm_info["hbm_headroom_gb_after_first_step"]
            scale = n_tok / 32768.0  # (savings quoted at cfg 3's 32 768 tokens per GPU)
            if short > 0:
                if eng.mem_level < 1 and short <= 19.8 * scale:
                    eng.mem_level = 1
                elif eng.mem_level < 2 and short <= 31.3 * scale:
                    eng.mem_level = 2
                else:
                    eng.save_activations = False
                    recompute_fallback = True
                torch.cuda.empty_cache()
            hbm_info["mem_level"] = eng.mem_level
    dev_sync()
    if dp:
        dist.barrier()
    dev_sync()
    if O is not None:
        O.profile_start()
    if sync is not None:
        sync.pop_timing()
        sync.timing = True
    if not dry:
        torch.cuda.reset_peak_memory_stats(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    dev_sync()
    if dp:
        dist.barrier()
    dev_sync()
    dt = time.perf_counter() - t0
    prof = O.profile_stop() if O is not None else {}
    prof_bytes = dict(O.LAST_PROFILE_BYTES) if O is not None else {}
    comm = None
    if sync is not None:
        comm = sync.pop_timing()
        sync.timing = False
    loss_val = float(loss.detach())
    peak_gb = 0.0 if dry else torch.cuda.max_memory_allocated(dev) / 1e9
    # ---- forward-only leg (same batch, same weights; not part of `value`) ----
    fwd_ms = None
    if not dry and not args.fwd_only and not args.no_forward_leg:
        nf = max(3, min(10, args.steps))
        with torch.no_grad():
            for _ in range(2):
                model(**dbatch)
            dev_sync()
            if dp:
                dist.barrier()
            tf0 = time.perf_counter()
            for _ in range(nf):
                model(**dbatch)
            dev_sync()
            if dp:
                dist.barrier()
            fwd_dt = time.perf_counter() - tf0
        fwd_ms = all_max(fwd_dt) / nf * 1e3
    dt = all_max(dt)
    if dry or args.fwd_only:
        n_tok_timed = n_tok * args.steps
    else:  # the positions the timed steps actually processed (step i of the run fed batch i mod n_rot; warm-up steps came first)
        n_tok_timed = sum(n_tok_rot[(args.warmup + i) % len(n_tok_rot)] for i in range(args.steps))
    n_tok_all = all_sum_int(n_tok_timed) / args.steps
    per_rank = None
    if dp:  # per-rank peak HBM and communication times, gathered on rank 0
        mine = torch.tensor([peak_gb, comm["comm_ms_total"] / args.steps, comm["comm_ms_exposed"] / args.steps], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[round(float(v), 2) for v in t] for t in allr]
    if rank != 0:
        if dp:
            dist.destroy_process_group()
        return
    ms = dt / args.steps * 1e3
    value = n_tok_all / (dt / args.steps)
    fwd = algorithmic_flops_fwd(B, S, n_img) if not dry else 0.0
    useful = fwd * (1.0 if args.fwd_only else 3.0)
    fp8_dom = "gemm_fp8" in prof and prof["gemm_fp8"][2] > prof.get("gemm_nt", (0, 0.0, 0.0))[2]
    n, work, gms = prof.get("gemm_fp8" if fp8_dom else "gemm_nt", (0, 0.0, 1e-9))
    ach = work / (gms * 1e-3) / 1e12
    peak = PEAK_FP8_TFLOPS if fp8_dom else PEAK_BF16_TFLOPS
    traffic, traffic_unit = read_traffic(args)
    step_desc = "fwd only" if args.fwd_only else ("fwd+bwd" + (" (layer recompute)" if not eng.save_activations else " (activations resident)") +
                                                 ("+allreduce" if dp else "") + "+clip+adamw")
    # FLOPs the step EXECUTED: every GEMM launch's 2*M*N*K as launched (event-timed profile of the timed steps: the scored-rows backward of the head
    # and the last layer contracts over ~15 % of the rows, which `useful` = 3 x forward still counts in full) + attention at the same 3 x convention
    att_fwd = 0.0 if dry else (B * 32 * 2.0 * S * S * 4096 + n_img * 23 * 4.0 * 577 * 577 * 1024)
    gemm_exec = (prof.get("gemm_nt", (0, 0.0, 0.0))[1] + prof.get("gemm_fp8", (0, 0.0, 0.0))[1]) / max(1, args.steps)
    executed = gemm_exec + att_fwd * (1.0 if args.fwd_only else 3.0)
    line = {
        "metric": "img-text tokens/sec/GPU (ViT-L + Llama-7B, 6-frame interpair, seq4096)",
        "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("cpu-fp32 (dry run)" if dry else "fp8-e4m3 decoder GEMMs (bf16 elsewhere)" if args.fp8_forward else
                  "fp8-e4m3 Linear GEMMs (decoder, CLIP tower, lm_head) fwd+dgrad+wgrad, per-row scales (bf16 residual stream / attention / norms, fp32 accumulate)" if args.fp8_train else
                  "bf16 (16-bit residual streams)" if args.residual_16bit else "bf16 (fp32 residual streams, fp32 accumulate)"),
        "data": "synthetic", "tokens_per_s_per_gpu": round(value / world, 1),
        "config": {"workload": workload, "per_gpu_batch": B, "seq_len": S, "images_per_gpu": n_img, "parallelism": f"dp{world}",
                   "step": step_desc, "loss": round(loss_val, 4),
                   # a real optimizer step shows as a falling loss (first warm-up step -> last timed step; with rotated batches of random token
                   # ids only what generalises falls: the unigram / position statistics, not the memorised batch)
                   "loss_first_warmup_step": (round(float(loss_first), 4) if loss_first is not None else None),
                   "batches": ("1 (the same batch every step: --same-batch)" if (dry or len(dbatches) == 1) else
                               f"{len(dbatches)} distinct seeded batches rotated through warm-up and the timed steps")},
        "useful_tflops_per_gpu": round(useful / (dt / args.steps) / 1e12, 1),
        # what the kernels ran (GEMM launches as launched + attention), and the step-level fraction on THAT basis; `useful` = 3 x algorithmic forward
        "executed_tflops_per_gpu": round(executed / (dt / args.steps) / 1e12, 1),
        "mfma_roofline_frac_step_useful": round(useful / (dt / args.steps) / 1e12 / PEAK_BF16_TFLOPS, 4),
        "peak_hbm_gb": round(peak_gb, 1),
        "hbm_headroom_gb": (None if dry else round(torch.cuda.get_device_properties(dev).total_memory / 1e9 - peak_gb, 1)),
        "mem_level": (None if dry else eng.mem_level),
        "mfma_roofline_frac_step": round(executed / (dt / args.steps) / 1e12 / PEAK_BF16_TFLOPS, 4),
        "roofline": {"kernel": ("gemm_nt_256<F8> (scaled-fp8 MFMA GEMM, all launches)" if fp8_dom else "bf16 MFMA GEMM kernels (gemm_nt_256 / gemm_w4 / gemm_nt_128, all launches)"),
                     "bound": "mfma", "achieved": round(ach, 1), "peak": peak,
                     "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic, "traffic_unit": traffic_unit, "launches": n,
                     "avg_launch_ms": round(gms / max(n, 1), 4), "gemm_share_of_step": round(gms / (dt * 1e3), 3),
                     "algorithmic_gb_per_launch": round(prof_bytes.get("gemm_fp8" if fp8_dom else "gemm_nt", 0.0) / max(n, 1) / 1e9, 4)},
    }
    if traffic:
        try:
            with open(traffic_summary_path()) as f:
                line["roofline"]["traffic_over_algorithmic"] = round(json.load(f)["traffic_over_algorithmic"], 2)
        except Exception:
            pass
    if dry:
        line["dry_run"] = True
    if dp:
        line["rccl_ranks"] = world
        line["comm_ms_total"] = round(max(r[1] for r in per_rank), 2)      # per step, slowest rank
        line["comm_ms_exposed"] = round(max(r[2] for r in per_rank), 2)    # per step: compute stream idle at the end of backward
        line["comm_collectives_per_step"] = comm["collectives"] // max(1, args.steps)
        line["comm_gb_per_step"] = round(comm["bytes"] / max(1, args.steps) / 1e9, 3)
        nb = line["comm_collectives_per_step"]
        if comm.get("each_ms") and nb and len(comm["each_ms"]) == nb * args.steps:  # rank 0's buckets, averaged over the timed steps, in issue order
            line["comm_per_bucket_ms"] = [round(sum(comm["each_ms"][i::nb]) / args.steps, 3) for i in range(nb)]
            line["comm_bucket_mb"] = [round(b / 1e6, 1) for b in comm.get("bucket_bytes", [])]
        # what RCCL was told (its own choice - ring / tree, LL / LL128 / simple - is per collective size unless pinned here)
        line["rccl_env"] = {k: os.environ.get(k, "default") for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS",
                                                                        "RCCL_MSCCL_ENABLE", "HSA_ENABLE_IPC_MODE_LEGACY")}
        line["rccl_choice"] = parse_rccl_tuning(locals().get("rccl_log"))
        # SURVEY 8e's budget for the step's gradient bytes on 7 x ~153 GB/s xGMI links per GPU: a ring keeps ONE link per direction busy
        # (2 (N-1)/N x bytes over one link), a direct reduce-scatter + all-gather spreads the same bytes over all N-1 peers' links
        gb, N_ = line["comm_gb_per_step"], world
        if N_ > 1:
            line["comm_expected_ms"] = {"ring_one_link": round(2 * (N_ - 1) / N_ * gb / 153.0 * 1e3, 1),
                                        "direct_all_links": round(2 * (N_ - 1) / N_ * gb / (153.0 * min(7, N_ - 1)) * 1e3, 1),
                                        "measured_total": line["comm_ms_total"], "measured_exposed": line["comm_ms_exposed"],
                                        "what": "per step; 14.09 GB at N = 8: 161 ms ring / 23 ms direct (SURVEY 8e)"}
        line["gemm_persistent"] = os.environ.get("MH_GEMM_PERSISTENT", "1") != "0"
        line["per_rank"] = [{"rank": i, "peak_hbm_gb": r[0], "comm_ms_total": r[1], "comm_ms_exposed": r[2]} for i, r in enumerate(per_rank)]
        line["recompute_fallback"] = recompute_fallback
        line.update(hbm_info)
    if not fp8_dom and not dry:
        sustained = sustained_mfma_tflops()
        if sustained:
            line["roofline"]["sustained_mfma_probe"] = {"tflops": round(sustained, 1), "frac_of_it": round(ach / sustained, 4),
                                                        "what": "register-only bf16 32x32x16 MFMA loop on pseudo-random operands, no memory traffic, this GPU, just now "
                                                                "(tools/probes/mfma_probe.hip): what the matrix pipes sustain under the power cap"}
    if fwd_ms is not None:
        line["forward_only"] = {"ms_per_step": round(fwd_ms, 2), "tokens_per_s_per_gpu": round(n_tok / (fwd_ms * 1e-3), 1),
                                "useful_tflops_per_gpu": round(fwd / (fwd_ms * 1e-3) / 1e12, 1),
                                "mfma_roofline_frac": round(fwd / (fwd_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)}
    if world == 1 and not dry and not args.no_extras and args.config == "cfg3" and not args.fwd_only and not args.fp8_train:
        try:  # BASELINE cfg 2 (single image + caption, S = 613, B = 1) on the same model: a few training steps + forwards
            b2 = synth.single_image_batch()
            d2 = to_dev(b2)
            S2 = int(b2["input_ids"].shape[1])
            for _ in range(2):
                train_step(d2)
            dev_sync()
            tq = time.perf_counter()
            for _ in range(10):
                train_step(d2)
            dev_sync()
            t_tr = (time.perf_counter() - tq) / 10
            with torch.no_grad():
                model(**d2)
                dev_sync()
                tq = time.perf_counter()
                for _ in range(10):
                    model(**d2)
                dev_sync()
                t_fw = (time.perf_counter() - tq) / 10
            f2 = algorithmic_flops_fwd(1, S2, 1)
            # At S = 613 the step is HBM-bound, not MFMA-bound: the roof is the bytes a step has to stream - forward: every 16-bit weight
            # once (14.09 GB); training step: weights twice (forward, dgrad) + 16-bit gradients written once + AdamW (read p, g, m, v; write p, m, v =
            # 22 B per parameter) - against 8 TB/s
            n_par = eng.arena.total
            b_fwd, b_train = 2.0 * n_par, (2.0 * 2 + 2.0 + 22.0) * n_par
            line["extras"] = {"cfg2": {"workload": "single 336px image + 32-token caption (S=613), B=1", "train_ms_per_step": round(t_tr * 1e3, 2),
                                       "train_tokens_per_s": round(S2 / t_tr, 1), "forward_ms": round(t_fw * 1e3, 2),
                                       "forward_tokens_per_s": round(S2 / t_fw, 1), "forward_mfma_roofline_frac": round(f2 / t_fw / 1e12 / PEAK_BF16_TFLOPS, 4),
                                       "roofline": {"bound": "hbm", "peak": 8000.0, "unit": "GB/s",
                                                    "forward": {"algorithmic_gb": round(b_fwd / 1e9, 2), "achieved": round(b_fwd / t_fw / 1e9, 1),
                                                                "frac": round(b_fwd / t_fw / 8e12, 4)},
                                                    "train_step": {"algorithmic_gb": round(b_train / 1e9, 2), "achieved": round(b_train / t_tr / 1e9, 1),
                                                                   "frac": round(b_train / t_tr / 8e12, 4)}}}}
        except Exception as e:
            line["extras"] = {"cfg2": f"failed: {e}"}
        if not args.no_cfg5_extra:
            line.setdefault("extras", {})["cfg5"] = cfg5_extra(model, eng, O, synth, to_dev, train_step, dev_sync, dev)
    if not dry and emu is not None:
        ws = [a_.elapsed_time(b_) for a_, b_ in emu["wait"][-args.steps:]]
        tot = args.warmup + args.steps
        line["comm_emulation"] = {"blocks": args.emulate_comm, "gb_streamed_per_step": round(emu["bytes"] / tot / 1e9, 2), "launch_pairs_per_step": emu["n"] // tot,
                                  "compute_stream_wait_ms_per_step": round(sum(ws) / max(1, len(ws)), 3),
                                  "what": "every gradient bucket read twice by a side-stream kernel of BLOCKS workgroups while the backward runs (stand-in for a ring all-reduce's kernels)"}
        eng.on_grads_ready = None
    if not args.no_cpu_baseline and world == 1 and not dry:  # (the reported CPU baseline belongs to the N = 1 line only)
        try:
            line["cpu_baseline"] = cpu_baseline()
        except Exception as e:  # the baseline leg must never take the GPU number down
            line["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port", "sample": f"failed: {e}"}
    os.write(result_fd, (json.dumps(line) + "\n").encode())
    if dp:
        dist.destroy_process_group()


def parse_rccl_tuning(path):
    """{message bytes: "ALGO/PROTO"} from RCCL's NCCL_DEBUG_SUBSYS=TUNING lines ("AllReduce: 404750336 Bytes -> Algo RING proto SIMPLE channel{Lo..Hi}={0..15}"); None when
    no such line was written (one-rank groups short-cut before the tuner; NCCL_DEBUG set by the caller; a library that words it differently)."""
    import re

    if not path or not os.path.exists(path):
        return None
    algo = {"0": "TREE", "1": "RING", "2": "COLLNET_DIRECT", "3": "COLLNET_CHAIN", "4": "NVLS", "5": "NVLS_TREE", "6": "PAT"}  # (numeric in older builds;
    proto = {"0": "LL", "1": "LL128", "2": "SIMPLE"}  # this image's librccl prints names: "%s: %ld Bytes -> Algo %s proto %s channel{Lo..Hi}={%d..%d}")
    out = {}
    try:
        with open(path, errors="replace") as f:
            for ln in f:
                m = re.search(r"AllReduce: (\d+) Bytes -> Algo (\S+) proto (\S+)(?: channel\{Lo\.\.Hi\}=\{(\d+)\.\.(\d+)\})?", ln)
                if m:
                    ch = f" ch{m.group(4)}-{m.group(5)}" if m.group(4) is not None else ""
                    out[int(m.group(1))] = f"{algo.get(m.group(2), m.group(2))}/{proto.get(m.group(3), m.group(3))}{ch}"
        os.remove(path)
    except Exception:
        return None
    return {str(k): v for k, v in sorted(out.items())} or None


def cfg5_extra(model, eng, O, synth, to_dev, train_step, dev_sync, dev, warm=2, steps=4):
    """BASELINE configs[4] on the same model, driver-timed like the rest of the line: S = 8192 interleave (4 images + long text per document), B = 4
    per GPU, every decoder and CLIP-tower Linear + lm_head on the scaled-fp8 MFMA (forward, dgrad, wgrad), bf16 elsewhere; a few training steps."""
    import gc

    try:
        gc.collect()
        torch.cuda.empty_cache()
        b5 = synth.interleave_batch(B=4, S=8192, n_images=4, rank=0)
        d5 = to_dev(b5)
        n_tok5 = int(b5["attention_mask"].sum())
        n_img5 = sum(int(im.shape[0]) for im in b5["images"])
        model.fp8_training = True
        tower_was, eng.fp8_tower = eng.fp8_tower, True  # (fp8 for all Llama / ViT Linear weights; +0.6 % against a 16-bit tower)
        torch.cuda.reset_peak_memory_stats(dev)
        for _ in range(warm):
            train_step(d5)
        dev_sync()
        O.profile_start()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = train_step(d5)
        dev_sync()
        dt = (time.perf_counter() - t0) / steps
        prof = O.profile_stop()
        n8, work8, ms8 = prof.get("gemm_fp8", (0, 0.0, 1e-9))
        n16, work16, ms16 = prof.get("gemm_nt", (0, 0.0, 1e-9))
        fwd = algorithmic_flops_fwd(4, 8192, n_img5)
        out = {"workload": "interleave (MMC4-style): B=4 x S=8192, 4 images per document, fp8 (e4m3) MFMA weight path (decoder + CLIP-tower Linears + "
                           "lm_head: forward, dgrad, wgrad), bf16 attention / norms, 16-bit residual streams",
               "ms_per_step": round(dt * 1e3, 2), "tokens_per_s": round(n_tok5 / dt, 1), "steps": steps, "warmup": warm,
               "loss": round(float(loss.detach()), 4), "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 1e9, 1),
               "useful_tflops": round(3 * fwd / dt / 1e12, 1), "parts_on_fp8": dict(getattr(eng, "last_fp8", {})),
               "roofline": {"kernel": "scaled-fp8 MFMA GEMMs (gemm_w4_f8 / gemm_nt_256<F8>, all launches)", "bound": "mfma",
                            "achieved": round(work8 / (ms8 * 1e-3) / 1e12, 1), "peak": PEAK_FP8_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(work8 / (ms8 * 1e-3) / 1e12 / PEAK_FP8_TFLOPS, 4), "launches": n8,
                            "share_of_step": round(ms8 / steps / (dt * 1e3), 3), "traffic": None,
                            "bf16_gemm_share_of_step": round(ms16 / steps / (dt * 1e3), 3)}}
        t5 = cfg5_traffic()
        if t5:
            out["roofline"].update(t5)
        return out
    except Exception as e:  # never take the headline down
        return f"failed: {type(e).__name__}: {e}"
    finally:
        model.fp8_training = False
        eng.fp8_tower = locals().get("tower_was", False)


def cfg5_traffic():
    """profiles/rNN_gemm_traffic_cfg5.json (tools/pmc_step_traffic.sh --config cfg5), quoted only when taken on this build's GEMM sources."""
    import glob

    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_gemm_traffic_cfg5.json")))
    if not found:
        return None
    try:
        with open(found[-1]) as f:
            t = json.load(f)
        if t.get("kernel_source_stamp") != kernel_source_stamp():
            return {"traffic_unit": f"profiles/{os.path.basename(found[-1])} was taken on other kernel sources: not quoted"}
        return {"traffic": round(t["traffic_bytes_per_launch"] / 1e9, 3), "traffic_over_algorithmic": round(t.get("traffic_over_algorithmic", 0), 2),
                "traffic_unit": f"GB per fp8 GEMM kernel launch (L2<->fabric, PMC: profiles/{os.path.basename(found[-1])})"}
    except Exception:
        return None


def sustained_mfma_tflops():
    """Median of three ~30 ms launches of the register-only MFMA loop (None if the probe library is not built)."""
    import ctypes

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "probes", "mfma_probe.so")
    if not os.path.exists(path):
        return None
    try:
        lib = ctypes.CDLL(path)
        sink = torch.zeros(1 << 16, dtype=torch.float32, device="cuda")
        blocks, iters, per_iter = 512, 100000, 8 * 32 * 32 * 16 * 2
        st = torch.cuda.current_stream().cuda_stream
        rates = []
        for i in range(4):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            lib.mfma_run(ctypes.c_void_p(sink.data_ptr()), blocks, iters, 32, 1, ctypes.c_void_p(st))
            e.record()
            torch.cuda.synchronize()
            if i:
                rates.append(blocks * 4 * iters * per_iter / (s.elapsed_time(e) * 1e-3) / 1e12)
        return sorted(rates)[1]
    except Exception:
        return None


if __name__ == "__main__":
    main()
