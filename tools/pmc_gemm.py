import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlin_amd import ops as O
dev = torch.device("cuda:0")
if os.environ.get("MH_GEMM_FORCE"):
    O.gemm_force_kernel(int(os.environ["MH_GEMM_FORCE"]))
T = 32768
a = torch.randn(T, 4096, device=dev).bfloat16(); w = torch.randn(12288, 4096, device=dev).bfloat16()
dy = torch.randn(T, 12288, device=dev).bfloat16()
out = torch.empty(T, 12288, device=dev, dtype=torch.bfloat16); dx = torch.empty(T, 4096, device=dev, dtype=torch.bfloat16); dw = torch.empty(12288, 4096, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    O.gemm_nt(a, w, out=out)                       # NT fwd
    O.gemm_nt(dy, w, b_t=True, out=dx)             # NN dgrad
    O.gemm_nt(dy, a, a_t=True, b_t=True, out=dw)   # TN wgrad
torch.cuda.synchronize()
