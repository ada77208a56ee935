"""A/B of the attention backward: dK + dV from one kernel (scores / dP computed once) vs the two single-output kernels.
Checks the outputs are bit-identical and times the backward both back-to-back ("cold": the chip idles at high clocks between
short launches) and interleaved with large GEMMs as in the training step ("hot": the power-capped clocks the step runs at)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlin_amd import ops as O

dev = torch.device("cuda:0")
A = torch.randn(16384, 4096, device=dev).to(torch.bfloat16)
Wt = torch.randn(8192, 4096, device=dev).to(torch.bfloat16)


def timeit(fn, iters=8, warm=3, hot=False):
    tot = 0.0
    for i in range(warm + iters):
        if hot:
            for _ in range(3):
                O.gemm_nt(A, Wt)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        if i >= warm:
            tot += s.elapsed_time(e)
    return tot / iters


MODES = [int(a) for a in sys.argv[1:]] or [0, 1]
for (B, S, H, D, causal, lens) in ((8, 4096, 32, 128, True, None), (4, 8192, 32, 128, True, None), (2, 4096, 32, 128, True, [4096, 3001]),
                                   (2, 1000, 8, 128, True, [1000, 517])):
    torch.manual_seed(0)
    qkv = torch.randn(B * S, 3 * H * D, device=dev).to(torch.bfloat16)
    q, k, v = (qkv[:, i * H * D:(i + 1) * H * D] for i in range(3))
    sl = torch.tensor(lens, dtype=torch.int32, device=dev) if lens else None
    o, lse = O.attn_fwd2(q, k, v, B, S, H, D, causal, seqlens=sl)
    do = torch.randn(B * S, H * D, device=dev).to(torch.bfloat16)
    fw = lambda: O.attn_fwd2(q, k, v, B, S, H, D, causal, seqlens=sl, out=o, lse=lse)
    print(f"   forward: cold {timeit(fw):.3f} hot {timeit(fw, hot=True):.3f} ms", flush=True)
    res, tc, th = {}, {}, {}
    for m in MODES + MODES:
        O.attn_bwd_fused_kv(m)
        res[m] = [x.clone() for x in O.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, causal, seqlens=sl)]
        run = lambda: O.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, causal, seqlens=sl)
        tc[m] = min(tc.get(m, 1e9), timeit(run))
        th[m] = min(th.get(m, 1e9), timeit(run, hot=True))
    O.attn_bwd_fused_kv(1)
    same = all(torch.equal(a, b) for m in MODES[1:] for a, b in zip(res[MODES[0]], res[m]))
    print(f"B={B} S={S} H={H} D={D} lens={lens}: " + "  ".join(f"mode {m}: cold {tc[m]:.3f} hot {th[m]:.3f} ms" for m in MODES) + f"  identical={same}", flush=True)
