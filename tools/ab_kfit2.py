"""t = a + b K fit of the 256-tile GEMM kernel (per-tile-round fixed cost and asymptotic rate): python tools/ab_kfit2.py [N]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from merlin_amd import ops as O
dev = torch.device("cuda:0")
M, N = 32768, (int(sys.argv[1]) if len(sys.argv) > 1 else 4096)
rounds = (M // 256) * (N // 256) / 256


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
for K in (1024, 2048, 4096, 8192):
    a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ts.append((K, timeit(lambda: O.gemm_nt(a, b, out=out))))
Ks = np.array([k for k, _ in ts], float); T = np.array([t for _, t in ts])
(a0, b0), *_ = np.linalg.lstsq(np.vstack([np.ones_like(Ks), Ks]).T, T, rcond=None)
print(" ".join(f"K={k}:{t*1e3:.3f}ms" for k, t in ts))
print(f"   fit: {a0*1e6/rounds:.2f} us per tile-round fixed; slope -> {2.0*M*N/b0/1e12:.0f} TFLOP/s asymptotic", flush=True)
