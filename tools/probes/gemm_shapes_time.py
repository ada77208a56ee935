"""Times the decoder's GEMM shapes (N(0,1) operands) with whichever library MH_LIB_PATH selects; prints TF/s and an output checksum (A/B arms must agree)."""
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
print("library:", os.environ.get("MH_LIB_PATH", "product"))
g = torch.Generator(device="cuda").manual_seed(7)
for mode, (M, N, K) in [("nt", (T, 12288, 4096)), ("nt", (T, 4096, 4096)), ("nt", (T, 22016, 4096)), ("nt", (T, 4096, 11008)), ("nn", (T, 4096, 12288)), ("nn", (T, 4096, 22016)), ("tn", (12288, 4096, T))]:
    if mode == "nt":
        a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16); b = torch.randn(N, K, device="cuda", generator=g).to(torch.bfloat16); kw = {}
    elif mode == "nn":
        a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16); b = torch.randn(K, N, device="cuda", generator=g).to(torch.bfloat16); kw = dict(b_t=True)
    else:
        a = torch.randn(K, M, device="cuda", generator=g).to(torch.bfloat16); b = torch.randn(K, N, device="cuda", generator=g).to(torch.bfloat16); kw = dict(a_t=True, b_t=True)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    t = min(timeit(lambda: O.gemm_nt(a, b, out=out, **kw)) for _ in range(3))
    print(f"{mode} M={M:5d} N={N:5d} K={K:5d}  {t:.4f} ms {2.0 * M * N * K / t / 1e9:6.0f} TF  checksum {float(out.float().abs().sum()):.6e}", flush=True)
