import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
T = 32768
for (M, N, K) in [(T, 12288, 4096), (T, 4096, 4096), (T, 22016, 4096), (T, 4096, 11008)]:
    a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16()
    for _ in range(5):
        c = a @ b.t()
    torch.cuda.synchronize()
