import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlin_amd import ops as O
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_ops import timeit
dev = torch.device("cuda:0")
T = 32768
for (M, N, K) in [(T, 12288, 4096), (T, 4096, 4096), (T, 22016, 4096), (T, 4096, 11008), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t16 = timeit(lambda: O.gemm_nt(a, b, out=out))
    qa, qb = O.quant_fp8_rows(a), O.quant_fp8_rows(b)
    t8 = timeit(lambda: O.gemm_fp8(qa, qb, out=out))
    tq = timeit(lambda: O.quant_fp8_rows(a))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: bf16 {t16*1e3:.3f} ms {fl/t16/1e12:.0f} TF | fp8 {t8*1e3:.3f} ms {fl/t8/1e12:.0f} TF | quant(A) {tq*1e3:.3f} ms", flush=True)
