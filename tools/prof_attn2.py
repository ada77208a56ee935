"""Tiny driver for PMC runs of the PRODUCT attention kernels: forward + backward at the cfg-3 geometry, causal and non-causal."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlin_amd import ops as O
dev = torch.device("cuda:0")
B, S, H, D = 8, 4096, 32, 128
causal = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
qkv = (torch.randn(B * S, 3 * H * D, device=dev) * 0.5).bfloat16()
q, k, v = (qkv[:, i * H * D:(i + 1) * H * D] for i in range(3))
do = torch.randn(B * S, H * D, device=dev).bfloat16()
dqkv = torch.empty_like(qkv)
form = int(os.environ.get("MH_ATTN_FWD_FORM", "0"))  # 0: attn_fwd2 (default), 1: attn_fwd3, 2: attn_fwd4
if form:
    O.attn_fwd_pingpong(form)
o, lse = O.attn_fwd2(q, k, v, B, S, H, D, causal=causal)
for _ in range(2):
    O.attn_fwd2(q, k, v, B, S, H, D, causal=causal, out=o, lse=lse)
    O.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, causal, dq=dqkv[:, :H * D], dk=dqkv[:, H * D:2 * H * D], dv=dqkv[:, 2 * H * D:])
if causal and os.environ.get("PMC_ALL_BWD_FORMS", "1") == "1":  # the seven-product form of the same backward (attn_bwd3_kv_k without the spill + attn_bwd2_dq_k) and the rounds 2-4 dK|dV kernel
    for _ in range(2):
        O.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, causal, dq=dqkv[:, :H * D], dk=dqkv[:, H * D:2 * H * D], dv=dqkv[:, 2 * H * D:], spill=False)
    O.attn_bwd_fused_kv(1)
    for _ in range(2):
        O.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, causal, dq=dqkv[:, :H * D], dk=dqkv[:, H * D:2 * H * D], dv=dqkv[:, 2 * H * D:], spill=False)
    O.attn_bwd_fused_kv(True)
torch.cuda.synchronize()
