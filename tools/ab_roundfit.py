"""Per-launch vs per-tile-round fixed cost of the 256-tile GEMM: t(rounds) at fixed K and N = 4096 (python tools/ab_roundfit.py [K])."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from merlin_amd import ops as O
dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = 4096


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
    return s.elapsed_time(e) / iters / 1e3


ts = []
for M in (4096, 8192, 16384, 32768, 65536):
    a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ts.append(((M // 256) * (N // 256) / 256, timeit(lambda: O.gemm_nt(a, b, out=out))))
R = np.array([r for r, _ in ts], float); T = np.array([t for _, t in ts])
(a0, b0), *_ = np.linalg.lstsq(np.vstack([np.ones_like(R), R]).T, T, rcond=None)
print(" ".join(f"rounds={r:g}:{t*1e6:.1f}us" for r, t in ts))
print(f"   K={K}: per launch {a0*1e6:.1f} us, per tile-round {b0*1e6:.2f} us (pure MFMA time of a round at 1.6 PF: {2*256*256*256*K/1.6e15*1e6:.2f} us)", flush=True)
