"""1-2 row GEMV at N = 4096: 2 vs 8 weight steps in flight per row (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlin_amd import ops as O

dev = torch.device("cuda:0")


def timeit(fn, iters=40, warm=5):
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


for (N, K) in ((4096, 4096), (4096, 11008), (8192, 4096), (12288, 4096)):
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    ws = [w.clone() for _ in range(max(1, int(600e6 // (N * K * 2))))]
    for M in (1, 2):
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        res = []
        for deep in (False, True):
            O.gemv_deep(deep)
            it = [0]

            def f():
                it[0] += 1
                O.gemv(x, ws[it[0] % len(ws)])
            t = timeit(f)
            res.append(f"{'deep' if deep else 'base'} {t * 1e3:6.1f} us {N * K * 2 / t / 1e9:5.2f} TB/s")
        print(f"N={N:5d} K={K:5d} M={M}: " + " | ".join(res), flush=True)
O.gemv_deep(True)
