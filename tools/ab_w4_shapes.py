"""A/B of the 4-wave (gemm_w4.hip, force 4) and 8-wave (gemm256.hip, force 256) 256x256 GEMM kernels on the training step's shapes,
interleaved rounds in one process (cdna guide rule 24).  usage: python tools/ab_w4_shapes.py [tn|nn|nt|all]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from merlin_amd import ops as O  # noqa: E402

dev = torch.device("cuda:0")
T, d, ff, V = 32768, 4096, 11008, 32064
which = sys.argv[1] if len(sys.argv) > 1 else "all"


def timeit(fn, n=6):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def rnd(*shape):
    return torch.randn(*shape, device=dev).bfloat16()


cases = []
if which in ("tn", "all"):
    for name, No, Ki in (("wgrad qkv", 3 * d, d), ("wgrad o", d, d), ("wgrad gu", 2 * ff, d), ("wgrad down", d, ff)):
        cases.append((name + f" TN [{No},{Ki},T]", "tn", No, Ki, T))
if which in ("nn", "all"):
    for name, N, K in (("dgrad qkv", d, 3 * d), ("dgrad o", d, d), ("dgrad gu", d, 2 * ff)):
        cases.append((name + f" NN [T,{N},{K}]", "nn", T, N, K))
if which in ("nt", "all"):
    for name, N, K in (("fwd o (plain)", d, d), ("fwd down (plain)", d, ff), ("fwd qkv (plain)", 3 * d, d)):
        cases.append((name + f" NT [T,{N},{K}]", "nt", T, N, K))
for name, form, M, N, K in cases:
    if form == "tn":
        a, b = rnd(K, M), rnd(K, N)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        fn = lambda: O.gemm_nt(a, b, a_t=True, b_t=True, out=out)  # noqa: E731
    elif form == "nn":
        a, b = rnd(M, K), rnd(K, N)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        fn = lambda: O.gemm_nt(a, b, b_t=True, out=out)  # noqa: E731
    else:
        a, b = rnd(M, K), rnd(N, K)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        fn = lambda: O.gemm_nt(a, b, out=out)  # noqa: E731
    res = {4: [], 256: []}
    for rd in range(3):
        for k in (256, 4):
            O.gemm_force_kernel(k)
            res[k].append(timeit(fn))
    O.gemm_force_kernel(0)
    fl = 2.0 * M * N * K
    m8, m4 = min(res[256]), min(res[4])
    print(f"{name:34s} 8-wave {m8:7.3f} ms {fl / m8 / 1e9:6.0f} TF | 4-wave {m4:7.3f} ms {fl / m4 / 1e9:6.0f} TF | {m8 / m4:5.3f}x", flush=True)
    del a, b, out
