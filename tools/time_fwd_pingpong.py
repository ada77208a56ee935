"""Time the ping-pong attention forward (csrc/attn_fwd3.hip) against attn_fwd2: python tools/time_fwd_pingpong.py  (MH_LIB_PATH selects a probe build, e.g. one compiled with -DPP_PROBE=n)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlin_amd import ops as O

H, D = 32, 128
for B, S, causal in ((4, 8192, False), (8, 4096, True)):
    qkv = (torch.randn(B * S, 3 * H * D, device="cuda") * 0.7).to(torch.bfloat16)
    q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
    for wide in (True, False):
        O.attn_fwd_pingpong(wide)
        for _ in range(3):
            O.attn_fwd2(q, k, v, B, S, H, D, causal)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            O.attn_fwd2(q, k, v, B, S, H, D, causal)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{os.environ.get('MH_LIB_PATH', 'product'):>28s} B={B} S={S} causal={causal} {'PING-PONG' if wide else 'fwd2'} {ms:.3f} ms  {4.0 * B * H * S * S * D * (0.5 if causal else 1.0) / ms / 1e9:.0f} TF", flush=True)
    O.attn_fwd_pingpong(False)
