"""Timing of the decoder's NT / NN products only (cfg-3 geometry) - the shape list the main-loop probes of gemm_w4.hip (-DW4H_EXPERIMENT=n, linked into a
second library selected with MH_LIB_PATH) are compared on; the probes' register-load forms do not honour the K-tail range check of the TN products.
    MH_LIB_PATH=tools/probes/libmerlin_w4h_e7.so python tools/probes/w4_probe_shapes.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from merlin_amd import ops as O

dev = torch.device("cuda:0")
if len(sys.argv) > 1:  # e.g. 256: the 8-wave kernel for everything (mh_gemm_force_kernel)
    O.gemm_force_kernel(int(sys.argv[1]))
T, d, ff = 32768, 4096, 11008
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


x, xf = rnd(T, d), rnd(T, ff)
wqkv, wo, wgu, wd = rnd(3 * d, d, scale=0.02), rnd(d, d, scale=0.02), rnd(2 * ff, d, scale=0.02), rnd(d, ff, scale=0.02)
rope = O.rope_table(4096, 128, 10000.0, dev)
x32 = torch.randn(T, d, device=dev)
dqkv, dgu = rnd(T, 3 * d), rnd(T, 2 * ff)
tot = 0.0
for name, fl, fn in [("fwd q|k|v + RoPE  NT [T,12288,4096]", 2.0 * T * 3 * d * d, lambda: O.gemm_nt_rope(x, wqkv, rope, 4096, 32, 128)),
                     ("fwd gate|up+SwiGLU NT [T,22016,4096]", 2.0 * T * 2 * ff * d, lambda: O.gemm_swiglu_fwd(x, wgu)),
                     ("fwd o -> fp32 += NT [T,4096,4096]", 2.0 * T * d * d, lambda: O.gemm_nt(x, wo, out=x32, accum=True)),
                     ("fwd down -> fp32 += NT [T,4096,11008]", 2.0 * T * d * ff, lambda: O.gemm_nt(xf, wd, out=x32, accum=True)),
                     ("dgrad qkv NN [T,4096,12288]", 2.0 * T * 3 * d * d, lambda: O.gemm_nt(dqkv, wqkv, b_t=True)),
                     ("dgrad gate|up NN [T,4096,22016]", 2.0 * T * 2 * ff * d, lambda: O.gemm_nt(dgu, wgu, b_t=True))]:
    ms = timeit(fn)
    tot += ms
    print(f"{name:42s} {ms:7.3f} ms {fl / ms / 1e9:6.0f} TF", flush=True)
print(f"sum {tot:.3f} ms")
