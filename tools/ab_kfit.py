import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlin_amd import ops as O
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_ops import timeit
dev = torch.device("cuda:0")
M, N = 32768, (int(sys.argv[1]) if len(sys.argv) > 1 else 4096)
tiles = (M // 256) * (N // 256)
rounds = tiles / 256
for which in (256, 4):
    ts = []
    for K in (1024, 2048, 4096, 8192):
        a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        if which == "lib":
            t = timeit(lambda: torch.matmul(a, b.t(), out=out))
        else:
            O.gemm_force_kernel(which)
            t = timeit(lambda: O.gemm_nt(a, b, out=out))
        ts.append((K, t))
    # least squares t = a + b*K
    import numpy as np
    Ks = np.array([k for k, _ in ts], float); T = np.array([t for _, t in ts])
    A = np.vstack([np.ones_like(Ks), Ks]).T
    (a0, b0), *_ = np.linalg.lstsq(A, T, rcond=None)
    print(which, " ".join(f"K={k}:{t*1e3:.3f}ms" for k, t in ts))
    print(f"   fit: overhead {a0*1e6:.1f} us total = {a0*1e6/rounds:.2f} us per tile-round; slope -> {2.0*M*N/b0/1e12:.0f} TFLOP/s asymptotic", flush=True)
O.gemm_force_kernel(0)
