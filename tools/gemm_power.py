"""GEMM throughput against operand data (zeros / ones / small integers / N(0,1)): the power cap, not the kernel, separates them (profiles/r03_gemm_power.txt)."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlin_amd import ops as O
T = 32768
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, M, N, K in (("fwd down NT", T, 4096, 11008), ("qkv NT", T, 12288, 4096)):
    for kind in ("randn", "zeros", "ones", "small-int"):
        if kind == "randn": a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
        elif kind == "zeros": a = torch.zeros(M, K, device="cuda").bfloat16(); b = torch.zeros(N, K, device="cuda").bfloat16()
        elif kind == "ones": a = torch.ones(M, K, device="cuda").bfloat16(); b = torch.ones(N, K, device="cuda").bfloat16()
        else: a = torch.randint(-2, 3, (M, K), device="cuda").bfloat16(); b = torch.randint(-2, 3, (N, K), device="cuda").bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for force in (256, 4):
            O.gemm_force_kernel(force)
            ms = t(lambda: O.gemm_nt(a, b, out=out))
            print(f"{name} {kind:9s} kernel {force:3d}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TF", flush=True)
        O.gemm_force_kernel(0)
