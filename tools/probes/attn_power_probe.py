"""Round 6: are the attention kernels POWER-limited or STALL-limited on this part?  Same kernels, same shapes (cfg 3: B = 8, S = 4096, H = 32, D = 128, causal),
same instruction streams - operands N(0,1) vs operands that do not toggle the data paths (all zeros; for the backward also dO = 0).  The guide's GEMM figure for the
same experiment is +19 % TF/s on zero-filled inputs (clock 2.30 vs 1.90-1.95 GHz).  A kernel that speeds up like that is bounded by the power cap (only fewer joules
per tile help); one that does not is bounded by its own issue / latency structure.
    python tools/probes/attn_power_probe.py            (on the GPU box)"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch  # noqa: E402

from merlin_amd import ops as O  # noqa: E402


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


B, S, H, D = 8, 4096, 32, 128
g = torch.Generator(device="cuda").manual_seed(1)
res = {}
for rep in range(2):
    for kind in ("random", "zeros", "random q,k / zero v,dO", "zero q,k / random v,dO"):
        qkv = torch.randn(B * S, 3 * H * D, generator=g, device="cuda").to(torch.bfloat16)
        do = torch.randn(B * S, H * D, generator=g, device="cuda").to(torch.bfloat16)
        if kind == "zeros":
            qkv.zero_(); do.zero_()
        elif kind.startswith("random q,k"):
            qkv[:, 2 * H * D:].zero_(); do.zero_()
        elif kind.startswith("zero q,k"):
            qkv[:, :2 * H * D].zero_()
        q, k, v = (qkv[:, i * H * D:(i + 1) * H * D] for i in range(3))
        o, lse = O.attn_fwd2(q, k, v, B, S, H, D, True)
        dq, dk, dv = O.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, True)
        tf = timeit(lambda: O.attn_fwd2(q, k, v, B, S, H, D, True, out=o, lse=lse))
        tb = timeit(lambda: O.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, True, dq=dq, dk=dk, dv=dv), iters=10, warm=3)
        res.setdefault(kind, []).append((tf, tb))
fl = 4.0 * B * H * S * S * D * 0.5
base = min(t[0] for t in res["random"]), min(t[1] for t in res["random"])
for kind, ts in res.items():
    tf, tb = min(t[0] for t in ts), min(t[1] for t in ts)
    print(f"{kind:26s} fwd {tf:.4f} ms {fl / tf / 1e9:5.0f} TF ({100 * (base[0] / tf - 1):+5.1f} %)   bwd (dK|dV + dQ + delta) {tb:.4f} ms {2.5 * fl / tb / 1e9:5.0f} TF ({100 * (base[1] / tb - 1):+5.1f} %)", flush=True)
# the GEMM for comparison (same experiment)
T, d = 32768, 4096
for kind in ("random", "zeros"):
    a = torch.randn(T, d, device="cuda").to(torch.bfloat16); w = (torch.randn(3 * d, d, device="cuda") * 0.02).to(torch.bfloat16)
    if kind == "zeros":
        a.zero_(); w.zero_()
    out = torch.empty(T, 3 * d, dtype=torch.bfloat16, device="cuda")
    t = min(timeit(lambda: O.gemm_nt(a, w, out=out), iters=10, warm=3) for _ in range(2))
    print(f"gemm NT [T,12288,4096] {kind:8s} {t:.4f} ms {2.0 * T * 3 * d * d / t / 1e9:5.0f} TF", flush=True)
