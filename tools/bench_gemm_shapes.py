"""Per-shape timing of every MFMA GEMM of one cfg-3 decoder layer (+ lm_head) in its three operand layouts, with the epilogues the
step uses, in isolation (run on the GPU box).  Shows which launches pull the step's average GEMM rate down.
Data ~ N(0, 1)-scaled like activations (the chip's clocks depend on the operand bits: zeros run faster)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from merlin_amd import ops as O

if "--no-persistent" in sys.argv:
    O.gemm_persistent(False)
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


rows = []


def rec(name, flops, fn):
    ms = timeit(fn)
    rows.append((name, ms, flops / ms / 1e9))
    print(f"{name:44s} {ms:8.3f} ms {flops / ms / 1e9:7.0f} TFLOP/s", flush=True)


x = rnd(T, d)
xf = rnd(T, ff)
wqkv, wo, wgu, wd, wlm = rnd(3 * d, d, scale=0.02), rnd(d, d, scale=0.02), rnd(2 * ff, d, scale=0.02), rnd(d, ff, scale=0.02), rnd(V, d, scale=0.02)
rope = O.rope_table(4096, 128, 10000.0, dev)
resid = rnd(T, d)
# forward (NT)
rec("fwd qkv+rope   NT [T,12288,4096]", 2 * T * 3 * d * d, lambda: O.gemm_nt_rope(x, wqkv, rope, 4096, 32, 128))
rec("fwd o+resid    NT [T,4096,4096]", 2 * T * d * d, lambda: O.gemm_nt(x, wo, resid=resid))
rec("fwd gu+swiglu  NT [T,22016,4096]", 2 * T * 2 * ff * d, lambda: O.gemm_swiglu_fwd(x, wgu))
rec("fwd down+resid NT [T,4096,11008]", 2 * T * d * ff, lambda: O.gemm_nt(xf, wd, resid=resid))
rec("fwd lm_head f32 NT [T,32064,4096]", 2 * T * V * d, lambda: O.gemm_nt(x, wlm, out_f32=True))
# dgrad (NN: weight read K-strided)
dqkv, dgu, gu = rnd(T, 3 * d), rnd(T, 2 * ff), rnd(T, 2 * ff)
rec("dgrad qkv      NN [T,4096,12288]", 2 * T * 3 * d * d, lambda: O.gemm_nt(dqkv, wqkv, b_t=True))
rec("dgrad o        NN [T,4096,4096]", 2 * T * d * d, lambda: O.gemm_nt(x, wo, b_t=True))
rec("dgrad gu       NN [T,4096,22016]", 2 * T * 2 * ff * d, lambda: O.gemm_nt(dgu, wgu, b_t=True))
rec("dgrad down+swiglu_bwd NN [T,11008,4096]", 2 * T * d * ff, lambda: O.gemm_swiglu_bwd(x, wd, gu))
dl = rnd(T, V)
rec("dgrad lm_head  NN [T,4096,32064]", 2 * T * V * d, lambda: O.gemm_nt(dl, wlm, b_t=True))
# wgrad (TN: both K-strided, contraction over tokens)
g1, g2, g3, g4, g5 = (torch.empty(3 * d, d, dtype=dt, device=dev), torch.empty(d, d, dtype=dt, device=dev), torch.empty(2 * ff, d, dtype=dt, device=dev),
                      torch.empty(d, ff, dtype=dt, device=dev), torch.empty(V, d, dtype=dt, device=dev))
rec("wgrad qkv      TN [12288,4096,T]", 2 * T * 3 * d * d, lambda: O.wgrad_tn(dqkv, x, g1, False))
rec("wgrad o        TN [4096,4096,T]", 2 * T * d * d, lambda: O.wgrad_tn(x, x, g2, False))
rec("wgrad gu       TN [22016,4096,T]", 2 * T * 2 * ff * d, lambda: O.wgrad_tn(dgu, x, g3, False))
rec("wgrad down     TN [4096,11008,T]", 2 * T * d * ff, lambda: O.wgrad_tn(x, xf, g4, False))
rec("wgrad lm_head  TN [32064,4096,T]", 2 * T * V * d, lambda: O.gemm_nt(dl, x, a_t=True, b_t=True, out=g5))
layer = sum(r[1] for r in rows if "lm_head" not in r[0])
print(f"per decoder layer: {layer:.2f} ms of GEMM -> x32 = {layer * 32:.0f} ms/step; lm_head {sum(r[1] for r in rows if 'lm_head' in r[0]):.2f} ms")
# wgrad with a K-contiguous activation operand (a transposed copy of x written by the producer): A K-strided, B K-contiguous
xT = x.t().contiguous()
xfT = xf.t().contiguous()
rec("wgrad qkv  A^T x^T-contig [12288,4096,T]", 2 * T * 3 * d * d, lambda: O.gemm_nt(dqkv, xT, a_t=True, out=g1))
rec("wgrad gu   A^T x^T-contig [22016,4096,T]", 2 * T * 2 * ff * d, lambda: O.gemm_nt(dgu, xT, a_t=True, out=g3))
rec("wgrad down A^T x^T-contig [4096,11008,T]", 2 * T * d * ff, lambda: O.gemm_nt(x, xfT, a_t=True, out=g4))
dqkvT = dqkv.t().contiguous()
rec("wgrad qkv  both contiguous NT [12288,4096,T]", 2 * T * 3 * d * d, lambda: O.gemm_nt(dqkvT, xT, out=g1))
