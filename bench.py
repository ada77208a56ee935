#!/usr/bin/env python
"""bench.py - the hot path's headline benchmark (BASELINE.json metric) on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3|cfg2] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

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


def cpu_baseline(seconds_budget=30.0):
    """Time the CPU oracle (oracle/ref_cpu.py, a port of the reference's CPU forward pinned to the
    reference by golden vectors) on the host cores, on a bounded sample of the cfg-3 workload:
    ONE sequence (S=4096, 6 frames): embed+splice, 2 of the 23 live ViT layers, 2 of the 32 decoder layers and
    lm_head+CE are timed (fp32, all host threads) and scaled to full depth."""
    from merlin_amd import synth
    from oracle import ref_cpu as R

    torch.manual_seed(0)
    nthreads = torch.get_num_threads()
    cfg = R.OracleConfig(num_hidden_layers=2, v_num_hidden_layers=3)  # select_layer=-2 -> 2 live ViT layers
    P = {k: torch.empty(s).normal_(0, 0.02) if len(s) > 1 else torch.ones(s) for k, s in R.param_shapes(cfg).items()}
    batch = synth.interpair_batch(B=1, S=4096)
    t_parts = {}
    with torch.no_grad():
        t0 = time.perf_counter()
        feats = R.encode_images(P, cfg, batch["images"])
        t_parts["vit2+proj"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        x = R.splice_image_features(P, cfg, batch["input_ids"], feats)
        t_parts["splice"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        h = R.llama_forward(P, cfg, x, batch["attention_mask"])
        t_parts["llama2"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        logits = torch.nn.functional.linear(h, P["lm_head.weight"])
        R.shifted_ce(logits, batch["labels"])
        t_parts["head"] = time.perf_counter() - t0
    full = t_parts["vit2+proj"] * 23 / 2 + t_parts["splice"] + t_parts["llama2"] * 32 / 2 + t_parts["head"]
    return {"value": round(4096 / full, 3), "unit": "tokens/s", "cores": nthreads, "kind": "port",
            "sample": "1 interpair sequence (S=4096, 6 frames) fp32 forward: 2/23 ViT layers + 2/32 decoder layers + lm_head+CE timed, "
                      f"scaled to full depth ({full:.1f} s/sequence est.; parts {json.dumps({k: round(v, 2) for k, v in t_parts.items()})})"}


def main():
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
    ap.add_argument("--fp8-forward", action="store_true", help="with --fwd-only: decoder Linears on the scaled-fp8 MFMA (e4m3 operands, "
                    "per-row scales); NOT the headline configuration (dtype field says so)")
    args = ap.parse_args()
    if os.environ.get("MH_GEMM_FORCE"):  # A/B arm selection for kernel development (see mh_gemm_force_kernel)
        from merlin_amd import ops as _O
        _O.gemm_force_kernel(int(os.environ["MH_GEMM_FORCE"]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from merlin_amd import ops as O
    from merlin_amd import synth
    from merlin_amd.dp import GradSync
    from merlin_amd.model.llama_mmgpt import build_synthetic_model
    from merlin_amd.optim import FusedAdamW

    assert O.arch_ok(local_rank), "bench.py needs a gfx950 (MI355X) device"
    model = build_synthetic_model(LLAMA_7B, VIT_L_336, projector="mlp", dtype=torch.bfloat16, device=dev, seed=0)
    model.engine.save_activations = not args.recompute
    if args.fp8_forward:
        assert args.fwd_only, "--fp8-forward is forward-only"
        model.fp8_forward = True
    if args.config == "cfg3":
        B = args.batch or 8
        batch = synth.interpair_batch(B=B, S=4096, rank=rank)
        workload = f"interpair: B={B}/GPU x S=4096 (6 x 336px frames + trajectory text), ViT-L/14-336 + mlp projector + Llama-7B"
    elif args.config == "cfg3-ragged":
        B = args.batch or 8
        batch = synth.interpair_batch(B=B, S=4096, rank=rank, ragged=True)
        workload = f"interpair ragged: B={B}/GPU, lengths <= 4096 right-padded (key-padding branch), 6 frames, ViT-L/14-336 + mlp + Llama-7B"
    elif args.config in ("cfg5", "cfg5-bf16"):
        B = args.batch or 4
        batch = synth.interleave_batch(B=B, S=8192, n_images=4, rank=rank)
        if args.config == "cfg5":
            args.fp8_train = True
            workload = f"interleave (MMC4-style): B={B}/GPU x S=8192, 4 images per document, fp8 (e4m3) MFMA weight path for the decoder's Linear layers"
        else:
            workload = f"interleave (MMC4-style): B={B}/GPU x S=8192, 4 images per document, bf16 weights (NOT cfg 5's fp8 weight path)"
    else:
        B = 1
        batch = synth.single_image_batch()
        workload = "single 336px image + 32-token caption (S=613), ViT-L/14-336 + mlp projector + Llama-7B"
    if args.fp8_train:
        model.fp8_training = True
    S = batch["input_ids"].shape[1]
    n_img = sum(int(im.shape[0]) for im in batch["images"])
    dbatch = dict(input_ids=batch["input_ids"].to(dev), attention_mask=batch["attention_mask"].to(dev),
                  labels=batch["labels"].to(dev), images=[im.to(dev) for im in batch["images"]])
    # the reference's recipe (pretrain.sh:23-29): --llrd, lr 5e-5, beta2 0.95, wd 0.05, cosine warm-up; HF's default
    # max_grad_norm=1.0 clipping - all of it inside the timed step
    from merlin_amd.optim import vit_lr_scale, cosine_with_warmup
    opt = FusedAdamW(model.engine, lr=5e-5, betas=(0.9, 0.95), weight_decay=0.05, lr_scale_fn=vit_lr_scale)
    it = [0]
    sync = GradSync(model.engine) if world > 1 else None

    def step():
        if args.fwd_only:
            with torch.no_grad():
                return model(**dbatch).loss
        out = model(**dbatch)
        out.loss.backward()
        # (GradSync joins the communication stream at the end of backward: the clip sees the all-reduced gradients)
        opt.step(grad_scale=(sync.grad_scale if sync else 1.0), max_grad_norm=1.0, lr_mult=cosine_with_warmup(it[0] + 10, 1000, 0.01))
        it[0] += 1
        opt.zero_grad()
        return out.loss

    for wi in range(args.warmup):
        tw = time.perf_counter()
        loss = step()
        if os.environ.get("MH_BENCH_PER_STEP"):  # warm-up profile (stderr): how many steps until the step time is flat
            torch.cuda.synchronize()
            print(f"[bench] warm-up step {wi}: {(time.perf_counter() - tw) * 1e3:.1f} ms", file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    O.profile_start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = O.profile_stop()
    loss_val = float(loss.detach())
    # ---- forward-only leg (same batch, same weights; not part of `value`) ----
    fwd_ms = None
    if not args.fwd_only and not args.no_forward_leg:
        nf = max(3, min(10, args.steps))
        with torch.no_grad():
            for _ in range(2):
                model(**dbatch)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            tf0 = time.perf_counter()
            for _ in range(nf):
                model(**dbatch)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            fwd_dt = time.perf_counter() - tf0
        if world > 1:
            t = torch.tensor([fwd_dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            fwd_dt = float(t)
        fwd_ms = fwd_dt / nf * 1e3
    n_tok = int(batch["attention_mask"].sum())  # the metric counts sequence positions, padding excluded
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
        c = torch.tensor([n_tok], device=dev, dtype=torch.int64)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        n_tok_all = int(c)
    else:
        n_tok_all = n_tok
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    ms = dt / args.steps * 1e3
    tokens = n_tok_all
    value = tokens / (dt / args.steps)
    fwd = algorithmic_flops_fwd(B, S, n_img)
    useful = fwd * (1.0 if args.fwd_only else 3.0)
    fp8_dom = "gemm_fp8" in prof and prof["gemm_fp8"][2] > prof.get("gemm_nt", (0, 0.0, 0.0))[2]
    n, work, gms = prof.get("gemm_fp8" if fp8_dom else "gemm_nt", (0, 0.0, 1e-9))
    ach = work / (gms * 1e-3) / 1e12
    peak = PEAK_FP8_TFLOPS if fp8_dom else PEAK_BF16_TFLOPS
    # HBM/fabric traffic of the dominant kernel comes from a separate rocprofv3 --pmc pass of this same command
    # (PMC collection cannot run inside the timed process); the committed summary is read back here.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r02_gemm_traffic.json")) as f:
            traffic = round(json.load(f)["traffic_bytes_per_launch"] / 1e9, 3) if args.config == "cfg3" and not args.fwd_only and not args.fp8_train else None
    except Exception:
        traffic = None
    line = {
        "metric": "img-text tokens/sec/GPU (ViT-L + Llama-7B, 6-frame interpair, seq4096)",
        "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": ("fp8-e4m3 decoder GEMMs (bf16 elsewhere)" if args.fp8_forward else
                                                                                          "fp8-e4m3 decoder GEMMs fwd+dgrad+wgrad, per-row scales (bf16 residual stream / attention / tower / head, fp32 accumulate)" if args.fp8_train else "bf16"),
        "data": "synthetic", "tokens_per_s_per_gpu": round(value / world, 1),
        "config": {"workload": workload, "per_gpu_batch": B, "seq_len": S, "images_per_gpu": n_img, "parallelism": f"dp{world}",
                   "step": "fwd only" if args.fwd_only else "fwd+bwd" + (" (layer recompute)" if args.recompute else " (activations resident)") + "+allreduce+adamw",
                   "loss": round(loss_val, 4)},
        "useful_tflops_per_gpu": round(useful / (dt / args.steps) / 1e12, 1),
        "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1),
        "mfma_roofline_frac_step": round(useful / (dt / args.steps) / 1e12 / PEAK_BF16_TFLOPS, 4),
        "roofline": {"kernel": ("gemm_nt_256<F8> (scaled-fp8 MFMA GEMM, all launches)" if fp8_dom else "gemm_nt_256/gemm_nt_128 (bf16 MFMA GEMM, all launches)"),
                     "bound": "mfma", "achieved": round(ach, 1), "peak": peak,
                     "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic, "traffic_unit": "GB per launch (L2<->fabric, PMC: profiles/r02_gemm_traffic.json)", "launches": n,
                     "avg_launch_ms": round(gms / max(n, 1), 4), "gemm_share_of_step": round(gms / (dt * 1e3), 3)},
    }
    if not fp8_dom:
        sustained = sustained_mfma_tflops()
        if sustained:
            line["roofline"]["sustained_mfma_probe"] = {"tflops": round(sustained, 1), "frac_of_it": round(ach / sustained, 4),
                                                        "what": "register-only bf16 32x32x16 MFMA loop on pseudo-random operands, no memory traffic, this GPU, just now "
                                                                "(tools/probes/mfma_probe.hip): what the matrix pipes sustain under the power cap"}
    if fwd_ms is not None:
        line["forward_only"] = {"ms_per_step": round(fwd_ms, 2), "tokens_per_s_per_gpu": round(n_tok / (fwd_ms * 1e-3), 1),
                                "useful_tflops_per_gpu": round(fwd / (fwd_ms * 1e-3) / 1e12, 1),
                                "mfma_roofline_frac": round(fwd / (fwd_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)}
    if not args.no_cpu_baseline and world == 1:  # (the reported CPU baseline belongs to the N = 1 line only)
        try:
            line["cpu_baseline"] = cpu_baseline()
        except Exception as e:  # the baseline leg must never take the GPU number down
            line["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port", "sample": f"failed: {e}"}
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


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
