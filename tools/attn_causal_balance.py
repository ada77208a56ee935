"""How much of the causal attention kernels' time is tile imbalance / diagonal overhead?  Times forward and backward at the cfg-3 and cfg-5
geometry causal vs NON-causal (twice the tiles, perfectly balanced blocks): a causal run at exactly half the non-causal time would be ideal."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlin_amd import ops as O

dev = torch.device("cuda:0")
H, D = 32, 128


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for B, S in ((8, 4096), (4, 8192)):
    qkv = (torch.randn(B * S, 3 * H * D, device=dev) * 0.5).bfloat16()
    q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
    do = torch.randn(B * S, H * D, device=dev).bfloat16()
    dqkv = torch.empty_like(qkv)
    for causal in (True, False):
        o, lse = O.attn_fwd2(q, k, v, B, S, H, D, causal=causal)
        O.attn_fwd_pingpong(True)
        tw = min(timeit(lambda: O.attn_fwd2(q, k, v, B, S, H, D, causal=causal, out=o, lse=lse)) for _ in range(2))
        O.attn_fwd_pingpong(False)
        print(f"B={B} S={S} causal={causal}: fwd PING-PONG {tw:.3f} ms ({4.0 * B * H * S * S * D * (0.5 if causal else 1.0) / tw / 1e9:.0f} TF)", flush=True)
        tf = timeit(lambda: O.attn_fwd2(q, k, v, B, S, H, D, causal=causal, out=o, lse=lse))
        tb = timeit(lambda: O.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, causal, dq=dqkv[:, :H * D], dk=dqkv[:, H * D:2 * H * D], dv=dqkv[:, 2 * H * D:]))
        fl = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
        print(f"B={B} S={S} causal={causal}: fwd {tf:.3f} ms ({fl / tf / 1e9:.0f} TF)  bwd {tb:.3f} ms ({2.5 * fl / tb / 1e9:.0f} TF)", flush=True)
