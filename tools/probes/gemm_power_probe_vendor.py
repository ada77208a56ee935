"""Round 6: where does the vendor NT kernel's 7-10 % come from - fewer joules per tile or fewer idle cycles?  The four decoder NT shapes, ours (gemm_w4) and the vendor
library (torch.matmul -> hipBLASLt; COMPARATOR ONLY), on N(0,1) operands and on operands that do not toggle the data paths (zeros).  On zeros the power cap is out of the
way and what is left is each kernel's own structure.
    python tools/probes/gemm_power_probe_vendor.py            (on the GPU box)"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch  # noqa: E402

from merlin_amd import ops as O  # noqa: E402


def timeit(fn, iters=10, warm=3):
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


T = 32768
print(f"{'NT shape':28s} {'ours random':>12s} {'ours zeros':>12s} {'ratio':>6s} | {'vendor random':>13s} {'vendor zeros':>12s} {'ratio':>6s} | vendor/ours random, zeros")
for (M, N, K) in [(T, 12288, 4096), (T, 4096, 4096), (T, 22016, 4096), (T, 4096, 11008)]:
    r = {}
    for kind in ("random", "zeros"):
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
        if kind == "zeros":
            a.zero_(); b.zero_()
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        mine = min(timeit(lambda: O.gemm_nt(a, b, out=out)) for _ in range(2))
        lib = min(timeit(lambda: torch.matmul(a, b.t(), out=out)) for _ in range(2))
        fl = 2.0 * M * N * K
        r[kind] = (fl / mine / 1e9, fl / lib / 1e9)
    (mr, vr), (mz, vz) = r["random"], r["zeros"]
    print(f"M={M} N={N:5d} K={K:5d}   {mr:9.0f} TF {mz:9.0f} TF {mr / mz:6.3f} | {vr:10.0f} TF {vz:9.0f} TF {vr / vz:6.3f} | {vr / mr:.3f}  {vz / mz:.3f}", flush=True)
