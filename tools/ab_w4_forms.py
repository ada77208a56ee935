"""Round 4: the 4-wave kernel's new forms against the 8-wave kernel, per shape, in isolation (cfg-3 geometry; N(0,1)-like operands).
    python tools/ab_w4_forms.py            (on the GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from merlin_amd import ops as O

dev = torch.device("cuda:0")
T, d, ff, V = 32768, 4096, 11008, 32064
dt = torch.bfloat16


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(dt)


def timeit(fn, iters=6, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def ab(name, flops, fn):
    res = {}
    for rnd_ in range(2):  # interleaved rounds
        for which in (256, 4):
            O.gemm_force_kernel(which)
            try:
                ms = timeit(fn)
            finally:
                O.gemm_force_kernel(0)
            res.setdefault(which, []).append(ms)
    a, b = min(res[256]), min(res[4])
    print(f"{name:46s} 8-wave {a:7.3f} ms {flops / a / 1e9:6.0f} TF | 4-wave {b:7.3f} ms {flops / b / 1e9:6.0f} TF | {100 * (a / b - 1):+5.1f} %", flush=True)


x, xf = rnd(T, d), rnd(T, ff)
wqkv, wo, wgu, wd, wlm = rnd(3 * d, d, scale=0.02), rnd(d, d, scale=0.02), rnd(2 * ff, d, scale=0.02), rnd(d, ff, scale=0.02), rnd(V, d, scale=0.02)
rope = O.rope_table(4096, 128, 10000.0, dev)
resid = rnd(T, d)
x32 = torch.randn(T, d, device=dev)
ab("fwd qkv + RoPE      NT [T,12288,4096]", 2 * T * 3 * d * d, lambda: O.gemm_nt_rope(x, wqkv, rope, 4096, 32, 128))
ab("fwd gate|up + SwiGLU NT [T,22016,4096]", 2 * T * 2 * ff * d, lambda: O.gemm_swiglu_fwd(x, wgu))
ab("fwd o + resid (16-bit) NT [T,4096,4096]", 2 * T * d * d, lambda: O.gemm_nt(x, wo, resid=resid))
ab("fwd o -> fp32 stream += NT [T,4096,4096]", 2 * T * d * d, lambda: O.gemm_nt(x, wo, out=x32, accum=True))
ab("fwd down -> fp32 stream += NT [T,4096,11008]", 2 * T * d * ff, lambda: O.gemm_nt(xf, wd, out=x32, accum=True))
lg = torch.empty(T, V, dtype=torch.float32, device=dev)
ab("fwd lm_head fp32 out NT [T,32064,4096]", 2 * T * V * d, lambda: O.gemm_nt(x, wlm, out=lg))
del lg
gu = rnd(T, 2 * ff)
ab("bwd down dgrad + SwiGLU' NN [T,11008,4096]", 2 * T * ff * d, lambda: O.gemm_swiglu_bwd(x, wd, gu))
del gu
# CLIP tower weight gradients: grouped launch vs one by one (each with its own split-K plan)
Tv, vd, vff = 48 * 577, 1024, 4096
shapes = [(vd, vff), (vff, vd), (vd, vd), (3 * vd, vd)]
probs = [(rnd(Tv, M, scale=0.5), rnd(Tv, N, scale=0.5), torch.zeros(M, N, dtype=dt, device=dev)) for M, N in shapes]
fl = sum(2.0 * M * N * Tv for M, N in shapes)


def one_by_one():
    for dy, xx, out in probs:
        O.wgrad_tn(dy, xx, out, accum=False)


for which, label in ((256, "8-wave split-K"), (0, "auto (4-wave split-K where it can)")):
    O.gemm_force_kernel(which)
    try:
        ms = timeit(one_by_one)
    finally:
        O.gemm_force_kernel(0)
    print(f"CLIP layer wgrads one by one, {label:36s} {ms:7.3f} ms {fl / ms / 1e9:6.0f} TF", flush=True)
ms = timeit(lambda: O.wgrad_tn_grouped(probs, accum=False, force=True))
print(f"CLIP layer wgrads GROUPED (192 tiles, one launch)                  {ms:7.3f} ms {fl / ms / 1e9:6.0f} TF", flush=True)
