"""Round 6: the raster group (tile rows one XCD visits together) of the 4-wave GEMM on the decoder's NT / NN shapes, N(0,1) operands: does operand traffic outside the CU
(L2 misses: 12 of 64 slice reads for a 4 x 8 concurrent set) cost speed on this power-capped part?    python tools/probes/gemm_raster_probe.py   (on the GPU box)"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch  # noqa: E402

from merlin_amd import ops as O  # noqa: E402


def timeit(fn, iters=10, warm=3):
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


T = 32768
gms = [1, 2, 4, 8, 16, 32]
print("TF/s per raster group " + " ".join(f"{g:>6d}" for g in gms))
for mode, (M, N, K) in [("nt", (T, 12288, 4096)), ("nt", (T, 4096, 4096)), ("nt", (T, 22016, 4096)), ("nt", (T, 4096, 11008)), ("nn", (T, 4096, 12288)), ("tn", (12288, 4096, T))]:
    if mode == "nt":
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
        kw = {}
    elif mode == "nn":
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = torch.randn(K, N, device="cuda").to(torch.bfloat16)
        kw = dict(b_t=True)
    else:
        a = torch.randn(K, M, device="cuda").to(torch.bfloat16); b = torch.randn(K, N, device="cuda").to(torch.bfloat16)
        kw = dict(a_t=True, b_t=True)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    res = {g: [] for g in gms}
    for _ in range(2):
        for g in gms:
            O.gemm_raster_group(g)
            res[g].append(timeit(lambda: O.gemm_nt(a, b, out=out, **kw)))
    O.gemm_raster_group(0)
    fl = 2.0 * M * N * K
    print(f"{mode} M={M:5d} N={N:5d} K={K:5d}  " + " ".join(f"{fl / min(res[g]) / 1e9:6.0f}" for g in gms), flush=True)
