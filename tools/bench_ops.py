"""Micro-benchmarks of the hot kernels at BASELINE cfg-3 shapes (run on the GPU box)."""
import sys, os, math, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "dev_arms")); import dev_ops as D  # needs MH_LIB_PATH=tools/dev_arms/libmerlin_hip_dev.so (python -m merlin_amd.csrc.build --dev)
from merlin_amd import ops as O

dev = torch.device("cuda:0")


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
    return s.elapsed_time(e) / iters * 1e-3


def bench_gemm(M, N, K, dtype=torch.bfloat16):
    a = torch.randn(M, K, device=dev).to(dtype)
    b = torch.randn(N, K, device=dev).to(dtype)
    out = torch.empty(M, N, device=dev, dtype=dtype)
    t = timeit(lambda: O.gemm_nt(a, b, out=out))
    tf = 2.0 * M * N * K / t / 1e12
    print(f"gemm M={M} N={N} K={K}: {t*1e3:.3f} ms  {tf:.0f} TFLOP/s  ({tf/2500*100:.1f}% of 2.5PF)", flush=True)
    return tf


def bench_attn(B, S, H, D, causal):
    dtype = torch.bfloat16
    qkv = torch.randn(B * S, 3 * H * D, device=dev).to(dtype)
    q, k, v = (qkv[:, i * H * D:(i + 1) * H * D] for i in range(3))
    vt = D.attn_prep_v(v, B, S, H, D)
    o, lse = D.attn_fwd(q, k, vt, B, S, H, D, causal)
    t = timeit(lambda: D.attn_fwd(q, k, vt, B, S, H, D, causal, out=o, lse=lse))
    fl = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
    print(f"attn_fwd B={B} S={S} H={H} D={D} causal={causal}: {t*1e3:.3f} ms  {fl/t/1e12:.0f} TFLOP/s", flush=True)
    if hasattr(O, "attn_fwd2"):
        t_ = timeit(lambda: O.attn_fwd2(q, k, v, B, S, H, D, causal, out=o, lse=lse))
        print(f"attn_fwd2 (no prep_v): {t_*1e3:.3f} ms  {fl/t_/1e12:.0f} TFLOP/s", flush=True)
    do = torch.randn(B * S, H * D, device=dev).to(dtype)
    t2 = timeit(lambda: D.attn_bwd(q, k, v, o, do, lse, B, S, H, D, causal), iters=5, warm=2)
    print(f"attn_bwd: {t2*1e3:.3f} ms  {2.5*fl/t2/1e12:.0f} TFLOP/s (5 matmuls counted)", flush=True)
    if hasattr(O, "attn_bwd2"):
        t_ = timeit(lambda: O.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, causal), iters=5, warm=2)
        print(f"attn_bwd2 (no re-layout passes): {t_*1e3:.3f} ms  {2.5*fl/t_/1e12:.0f} TFLOP/s", flush=True)
    t3 = timeit(lambda: D.attn_prep_v(v, B, S, H, D, out=vt))
    print(f"prep_v: {t3*1e3:.3f} ms  {2*B*S*H*D*2/t3/1e9:.0f} GB/s", flush=True)


if __name__ == "__main__":
    T = 32768
    if "attn" in sys.argv:
        bench_attn(8, 4096, 32, 128, True)
        bench_attn(48, 577, 16, 64, False)
        sys.exit(0)
    for (M, N, K) in [(T, 12288, 4096), (T, 4096, 4096), (T, 22016, 4096), (T, 4096, 11008), (T, 32064, 4096),
                      (4096, 4096, 4096), (8192, 8192, 8192), (613, 12288, 4096), (27696, 3072, 1024), (27696, 4096, 1024), (27696, 1024, 4096)]:
        bench_gemm(M, N, K)
    bench_attn(8, 4096, 32, 128, True)
    bench_attn(48, 577, 16, 64, False)
    x = torch.randn(T, 4096, device=dev).bfloat16(); w = torch.ones(4096, device=dev).bfloat16(); y = torch.empty_like(x)
    t = timeit(lambda: O.rmsnorm_fwd(x, w, 1e-6, out=y)); print(f"rmsnorm_fwd: {t*1e3:.3f} ms {2*x.numel()*2/t/1e9:.0f} GB/s")
    gu = torch.randn(T, 22016, device=dev).bfloat16(); act = torch.empty(T, 11008, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: O.swiglu_fwd(gu, out=act)); print(f"swiglu_fwd: {t*1e3:.3f} ms {(gu.numel()+act.numel())*2/t/1e9:.0f} GB/s")
    tr = torch.empty(4096, T, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: O.transpose16(x, out=tr)); print(f"transpose16: {t*1e3:.3f} ms {2*x.numel()*2/t/1e9:.0f} GB/s")
