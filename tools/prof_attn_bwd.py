"""Runs the causal D = 128 backward in its three forms (tools/ab_attn_kv3.py) a few times each, for `rocprofv3 --kernel-trace --stats`:
per-kernel durations of attn_bwd2_kv_k<MODE 3> / attn_bwd3_kv_k<spill off | on> / attn_bwd2_dq_k / attn_bwd3_dq_k at cfg 3 and cfg 5."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa: E402

from merlin_amd import _lib as L  # noqa: E402
from merlin_amd import ops as O  # noqa: E402

H, D = 32, 128
for B, S in ((8, 4096), (4, 8192)):
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = torch.randn(B * S, 3 * H * D, generator=g, device="cuda").to(torch.bfloat16)
    do = torch.randn(B * S, H * D, generator=g, device="cuda").to(torch.bfloat16)
    q, k, v = (qkv[:, i * H * D:(i + 1) * H * D] for i in range(3))
    o, lse = O.attn_fwd2(q, k, v, B, S, H, D, True)
    dq, dk, dv = (torch.empty_like(o) for _ in range(3))
    for rep in range(3):
        for m in (1, 2, 3):
            L.lib().mh_attn_bwd_fused_kv(C.c_int(min(m, 2)))
            for _ in range(5):
                O.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, True, dq=dq, dk=dk, dv=dv, spill=(m == 3))
    torch.cuda.synchronize()
    del qkv, do, o, lse, dq, dk, dv
