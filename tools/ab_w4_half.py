"""Round 5: 128-row block tiles of the 4-wave GEMM (gemm_w4<..., MI = 4>) against its 256-row tiles on the NT products of a short prefill
(cfg 2: one 613-token sequence; Llama-7B widths), in isolation, weights streamed from HBM (a different weight per call: 32 layers).
    python tools/ab_w4_half.py [M ...]            (on the GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from merlin_amd import ops as O

dev = torch.device("cuda:0")
d, ff, V, L = 4096, 11008, 32064, 8
dt = torch.bfloat16


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(dt)


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters / L * 1e3  # us per call


wqkv = [rnd(3 * d, d, scale=0.02) for _ in range(L)]
wgu = [rnd(2 * ff, d, scale=0.02) for _ in range(L)]
wo = [rnd(d, d, scale=0.02) for _ in range(L)]
wlm = rnd(V, d, scale=0.02)
rope = O.rope_table(4096, 128, 10000.0, dev)
for M in [int(a) for a in sys.argv[1:]] or [613, 101, 128, 256, 384, 1024, 2048]:
    x = rnd(M, d)
    x32 = torch.randn(M, d, device=dev)
    lg = torch.empty(M, V, dtype=torch.float32, device=dev)
    forms = [("q|k|v + RoPE  [M,12288,4096]", 2.0 * M * 3 * d * d, lambda: [O.gemm_nt_rope(x, w, rope, 4096, 32, 128) for w in wqkv]),
             ("gate|up + SwiGLU [M,22016,4096]", 2.0 * M * 2 * ff * d, lambda: [O.gemm_swiglu_fwd(x, w) for w in wgu]),
             ("o -> fp32 stream += [M,4096,4096] (one pass)", 2.0 * M * d * d, lambda: [O.L.check(O.L.lib().mh_gemm(O.p(x), O.i64(d), O.i32(0), O.p(w), O.i64(d), O.i32(0), O.p(x32), O.i64(d), None, None, O.i64(0), O.i32(M), O.i32(d), O.i32(d), O.i32(O.dt_of(x)), O.i32(O.EPI_OUT_F32 | O.EPI_ACCUM), O._stream()), "mh_gemm") for w in wo]),
             ("lm_head fp32 logits [M,32064,4096]", 2.0 * M * V * d, lambda: [O.gemm_nt(x, wlm, out=lg) for _ in range(L)])]
    for name, fl, fn in forms:
        res = {}
        for _ in range(2):
            for mode in (0, 1, 2):
                O.gemm_w4_half(mode)
                try:
                    res.setdefault(mode, []).append(timeit(fn))
                finally:
                    O.gemm_w4_half(1)
        a, b, c = min(res[0]), min(res[1]), min(res[2])
        print(f"M={M:5d} {name:46s} 256-row {a:7.1f} us {fl / a / 1e6:6.0f} TF | auto {b:7.1f} us | 128-row {c:7.1f} us {fl / c / 1e6:6.0f} TF | {100 * (a / c - 1):+5.1f} %", flush=True)
