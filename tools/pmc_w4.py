import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlin_amd import ops as O
dev = torch.device("cuda:0")
T = 32768
a = torch.randn(T, 4096, device=dev).bfloat16(); w = torch.randn(12288, 4096, device=dev).bfloat16()
out = torch.empty(T, 12288, device=dev, dtype=torch.bfloat16)
for which in (256, 4):
    O.gemm_force_kernel(which)
    for _ in range(2):
        O.gemm_nt(a, w, out=out)
for _ in range(2):
    torch.matmul(a, w.t(), out=out)
torch.cuda.synchronize()
