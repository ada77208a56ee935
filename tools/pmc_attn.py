"""Tiny driver for PMC runs: one attention fwd + bwd at cfg-3 shape (few launches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "dev_arms")); import dev_ops as D  # needs MH_LIB_PATH=tools/dev_arms/libmerlin_hip_dev.so (python -m merlin_amd.csrc.build --dev)
from merlin_amd import ops as O
dev = torch.device("cuda:0")
B, S, H, D = 8, 4096, 32, 128
qkv = torch.randn(B * S, 3 * H * D, device=dev).bfloat16()
q, k, v = (qkv[:, i * H * D:(i + 1) * H * D] for i in range(3))
vt = D.attn_prep_v(v, B, S, H, D)
o, lse = D.attn_fwd(q, k, vt, B, S, H, D, True)
do = torch.randn(B * S, H * D, device=dev).bfloat16()
for _ in range(2):
    D.attn_fwd(q, k, vt, B, S, H, D, True, out=o, lse=lse)
    D.attn_bwd(q, k, v, o, do, lse, B, S, H, D, True)
torch.cuda.synchronize()
