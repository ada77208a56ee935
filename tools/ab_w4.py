import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlin_amd import ops as O
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_ops import timeit
dev = torch.device("cuda:0")
T = 32768
shapes = [(T, 12288, 4096, ""), (T, 4096, 4096, "r"), (T, 22016, 4096, ""), (T, 4096, 11008, "r"), (T, 32064, 4096, "f"),
          (27696, 3072, 1024, "b"), (27696, 1024, 1024, "br"), (27696, 4096, 1024, "bg"), (27696, 1024, 4096, "br")]
for (M, N, K, fl) in shapes:
    a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16()
    bias = torch.randn(N, device=dev).bfloat16() if "b" in fl else None
    resid = torch.randn(M, N, device=dev).bfloat16() if "r" in fl else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if "f" in fl else torch.bfloat16)
    res = []
    for which in (256, 88, 256, 88):
        O.gemm_force_kernel(which)
        t = timeit(lambda: O.gemm_nt(a, b, out=out, bias=bias, resid=resid, act="quick_gelu" if "g" in fl else None, out_f32="f" in fl))
        res.append(f"{which}: {t*1e3:.3f} ms {2.0*M*N*K/t/1e12:.0f} TF")
    print(f"M={M} N={N} K={K} [{fl}]  " + " | ".join(res), flush=True)
O.gemm_force_kernel(0)
