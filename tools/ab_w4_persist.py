"""Round 6: the PERSISTENT form of the 4-wave GEMM (gemm_w4p: one block per CU walking its output tiles, the next tile's first two K-tiles requested under the
finished tile's global stores) against the one-block-per-tile form, per decoder product of the cfg-3 step, in isolation: bitwise equality of every output and
time per call, interleaved rounds.
    python tools/ab_w4_persist.py            (on the GPU box)"""
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


def timeit(fn, iters=8, warm=3):
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


def ab(name, flops, fn, outs):
    """fn() runs the product and returns nothing; outs() returns the tensors it wrote (cloned for the comparison)."""
    res, snap = {0: [], 1: []}, {}
    for rnd_ in range(3):
        for on in (0, 1):
            O.gemm_w4_persistent(on)
            try:
                if rnd_ == 0:
                    fn()
                    torch.cuda.synchronize()
                    snap[on] = [t.clone() for t in outs()]
                res[on].append(timeit(fn))
            finally:
                O.gemm_w4_persistent(1)
    same = all(torch.equal(a, b) for a, b in zip(snap[0], snap[1]))
    a, b = min(res[0]), min(res[1])
    print(f"{name:50s} per tile {a:7.3f} ms {flops / a / 1e9:6.0f} TF | persistent {b:7.3f} ms {flops / b / 1e9:6.0f} TF | {100 * (a / b - 1):+5.1f} % | bitwise equal: {same}", flush=True)
    return same


ok = True
x, xf = rnd(T, d), rnd(T, ff)
wqkv, wo, wgu, wd, wlm = rnd(3 * d, d, scale=0.02), rnd(d, d, scale=0.02), rnd(2 * ff, d, scale=0.02), rnd(d, ff, scale=0.02), rnd(V, d, scale=0.02)
rope = O.rope_table(4096, 128, 10000.0, dev)
resid = rnd(T, d)
hold = {}


def run(key, f):
    hold[key] = f()


ok &= ab("fwd q|k|v + RoPE         NT [T,12288,4096]", 2 * T * 3 * d * d, lambda: run("a", lambda: O.gemm_nt_rope(x, wqkv, rope, 4096, 32, 128)), lambda: [hold["a"]])
ok &= ab("fwd gate|up + SwiGLU     NT [T,22016,4096]", 2 * T * 2 * ff * d, lambda: run("b", lambda: O.gemm_swiglu_fwd(x, wgu)), lambda: list(hold["b"]))
hold.clear()
ok &= ab("fwd o + resid (16-bit)   NT [T,4096,4096]", 2 * T * d * d, lambda: run("c", lambda: O.gemm_nt(x, wo, resid=resid)), lambda: [hold["c"]])
x32 = torch.zeros(T, d, device=dev)


def acc_o():
    x32.zero_()
    O.gemm_nt(x, wo, out=x32, accum=True)


def acc_d():
    x32.zero_()
    O.gemm_nt(xf, wd, out=x32, accum=True)


ok &= ab("fwd o -> fp32 stream +=  NT [T,4096,4096]  (+ zero fill)", 2 * T * d * d, acc_o, lambda: [x32])
ok &= ab("fwd down -> fp32 stream += NT [T,4096,11008] (+ zero fill)", 2 * T * d * ff, acc_d, lambda: [x32])
lg = torch.empty(T, V, dtype=torch.float32, device=dev)
ok &= ab("fwd lm_head fp32 out     NT [T,32064,4096]", 2 * T * V * d, lambda: O.gemm_nt(x, wlm, out=lg), lambda: [lg])
del lg
hold.clear()
gu = rnd(T, 2 * ff)
ok &= ab("bwd down dgrad + SwiGLU' NN [T,11008,4096]", 2 * T * ff * d, lambda: run("e", lambda: O.gemm_swiglu_bwd(x, wd, gu)), lambda: [hold["e"]])
del gu
hold.clear()
dqkv, dgu = rnd(T, 3 * d), rnd(T, 2 * ff)
o1, o2, o3 = torch.empty(T, d, dtype=dt, device=dev), torch.empty(T, d, dtype=dt, device=dev), torch.empty(T, d, dtype=dt, device=dev)
ok &= ab("bwd q|k|v dgrad          NN [T,4096,12288]", 2 * T * d * 3 * d, lambda: O.gemm_nt(dqkv, wqkv, out=o1, b_t=True), lambda: [o1])
ok &= ab("bwd o dgrad              NN [T,4096,4096]", 2 * T * d * d, lambda: O.gemm_nt(x, wo, out=o2, b_t=True), lambda: [o2])
ok &= ab("bwd gate|up dgrad        NN [T,4096,22016]", 2 * T * d * 2 * ff, lambda: O.gemm_nt(dgu, wgu, out=o3, b_t=True), lambda: [o3])
# ragged row counts (partial last tile row, tiles > CUs) and fp16
for M in (32768 - 100, 20000):
    xa = rnd(M, d)
    oo = torch.empty(M, 3 * d, dtype=dt, device=dev)
    ok &= ab(f"plain NT [M={M},12288,4096]", 2 * M * 3 * d * d, lambda: O.gemm_nt(xa, wqkv, out=oo), lambda: [oo])
xh, wh = x.to(torch.float16), wqkv.to(torch.float16)
oh = torch.empty(T, 3 * d, dtype=torch.float16, device=dev)
ok &= ab("plain NT fp16 [T,12288,4096]", 2 * T * 3 * d * d, lambda: O.gemm_nt(xh, wh, out=oh), lambda: [oh])
print("ALL BITWISE EQUAL" if ok else "MISMATCH", flush=True)
