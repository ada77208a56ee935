"""Yardstick (NOT on the product path): time the vendor libraries (hipBLASLt via torch.matmul, torch SDPA) on the
step's shapes next to the hand-written kernels, to see how much headroom the kernels have on this chip."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from merlin_amd import ops as O
from bench_ops import timeit

dev = torch.device("cuda:0")
T = 32768


def gemm(M, N, K, mode):
    dt = torch.bfloat16
    if mode == "nt":   # fwd: out[M,N] = a[M,K] w[N,K]^T
        a = torch.randn(M, K, device=dev).to(dt); b = torch.randn(N, K, device=dev).to(dt)
        mine = lambda: O.gemm_nt(a, b, out=out)
        lib = lambda: torch.matmul(a, b.t(), out=out)
    elif mode == "nn":  # dgrad: out[M,N] = a[M,K] w[K,N]
        a = torch.randn(M, K, device=dev).to(dt); b = torch.randn(K, N, device=dev).to(dt)
        mine = lambda: O.gemm_nt(a, b, out=out, b_t=True)
        lib = lambda: torch.matmul(a, b, out=out)
    else:               # wgrad: out[M,N] = a[K,M]^T b[K,N]
        a = torch.randn(K, M, device=dev).to(dt); b = torch.randn(K, N, device=dev).to(dt)
        mine = lambda: O.gemm_nt(a, b, out=out, a_t=True, b_t=True)
        lib = lambda: torch.matmul(a.t(), b, out=out)
    out = torch.empty(M, N, device=dev, dtype=dt)
    t1 = timeit(mine); t2 = timeit(lib)
    fl = 2.0 * M * N * K
    print(f"gemm {mode} M={M} N={N} K={K}: mine {t1*1e3:.3f} ms {fl/t1/1e12:.0f} TF | hipblaslt {t2*1e3:.3f} ms {fl/t2/1e12:.0f} TF", flush=True)


def attn(B, S, H, D, causal):
    dt = torch.bfloat16
    q, k, v = (torch.randn(B, H, S, D, device=dev).to(dt).requires_grad_(True) for _ in range(3))
    fl = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
    for name, be in (("flash", torch.nn.attention.SDPBackend.FLASH_ATTENTION), ("efficient", torch.nn.attention.SDPBackend.EFFICIENT_ATTENTION)):
        try:
            with torch.nn.attention.sdpa_kernel(be):
                o = F.scaled_dot_product_attention(q, k, v, is_causal=causal)
                do = torch.randn_like(o)
                t1 = timeit(lambda: F.scaled_dot_product_attention(q, k, v, is_causal=causal))
                def fb():
                    o = F.scaled_dot_product_attention(q, k, v, is_causal=causal)
                    o.backward(do)
                t2 = timeit(fb, iters=5, warm=2)
            print(f"sdpa[{name}] B={B} S={S} H={H} D={D} causal={causal}: fwd {t1*1e3:.3f} ms {fl/t1/1e12:.0f} TF | fwd+bwd {t2*1e3:.3f} ms (bwd ~{(t2-t1)*1e3:.3f} ms {2.5*fl/(t2-t1)/1e12:.0f} TF)", flush=True)
        except Exception as e:  # noqa
            print(f"sdpa[{name}] unavailable: {str(e)[:100]}")


if __name__ == "__main__":
    for (M, N, K) in [(T, 12288, 4096), (T, 4096, 4096), (T, 22016, 4096), (T, 4096, 11008), (T, 32064, 4096), (27696, 3072, 1024), (27696, 4096, 1024), (27696, 1024, 4096)]:
        gemm(M, N, K, "nt")
    for (M, N, K) in [(T, 4096, 12288), (T, 4096, 4096), (T, 4096, 22016), (T, 11008, 4096)]:
        gemm(M, N, K, "nn")
    for (M, N, K) in [(12288, 4096, T), (4096, 4096, T), (22016, 4096, T), (4096, 11008, T)]:
        gemm(M, N, K, "tn")
    attn(8, 4096, 32, 128, True)
    attn(48, 577, 16, 64, False)
