import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlin_amd import ops as O
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_ops as B
T = 32768
a = torch.randn(1000, 256, device="cuda").bfloat16(); b = torch.randn(515, 256, device="cuda").bfloat16()
O.gemm_force_kernel(256); r1 = O.gemm_nt(a, b, out_f32=True); O.gemm_force_kernel(4); r2 = O.gemm_nt(a, b, out_f32=True)
print("w4 vs 256 max rel diff", float((r1 - r2).abs().max() / r1.abs().max()))
for rnd in range(2):
    for which in (256, 4):
        O.gemm_force_kernel(which); print("kernel", which)
        for (M, N, K) in [(T, 12288, 4096), (T, 4096, 4096), (T, 22016, 4096), (T, 4096, 11008), (T, 32064, 4096), (27696, 4096, 1024), (27696, 1024, 4096), (8192, 8192, 8192)]:
            B.bench_gemm(M, N, K)
O.gemm_force_kernel(0)
