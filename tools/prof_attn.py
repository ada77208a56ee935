"""One attention forward + backward geometry under rocprofv3 (per-kernel times, SQ counters): python tools/prof_attn.py [unused] [S] [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from merlin_amd import ops as O

S = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
H, D = 32, 128
dev = torch.device("cuda:0")
qkv = torch.randn(B * S, 3 * H * D, device=dev).bfloat16()
q, k, v = (qkv[:, i * H * D:(i + 1) * H * D] for i in range(3))
do = torch.randn(B * S, H * D, device=dev).bfloat16()
o, lse = O.attn_fwd2(q, k, v, B, S, H, D, True)
dq, dk, dv = O.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, True)
for _ in range(6):
    O.attn_fwd2(q, k, v, B, S, H, D, True, out=o, lse=lse)
    O.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, True, dq=dq, dk=dk, dv=dv)
torch.cuda.synchronize()
