"""The vision tower's GEMM shapes (48 frames x 577 tokens = 27 696 rows, width 1024) in their three layouts, 256^2 vs 128^2 tiles (run on the GPU box).
The wgrad column is the PLAIN TN product (16-64 output tiles for 256 CUs): the engine runs these weight gradients as split-K launches
(ops.gemm_splitk + tail plans), so only the fwd / dgrad columns say something about the step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlin_amd import ops as O
dev = torch.device("cuda:0")
R = 27696


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def rnd(*s):
    return torch.randn(*s, device=dev).bfloat16()


tot = {0: 0.0, 128: 0.0, 256: 0.0}
for (N, K, name) in ((3072, 1024, "qkv"), (1024, 1024, "out"), (4096, 1024, "fc1"), (1024, 4096, "fc2")):
    x, w, dy = rnd(R, K), rnd(N, K), rnd(R, N)
    y, dx, dw = torch.empty(R, N, device=dev, dtype=torch.bfloat16), torch.empty(R, K, device=dev, dtype=torch.bfloat16), torch.empty(N, K, device=dev, dtype=torch.bfloat16)
    row = f"{name:4s} N={N:5d} K={K:5d}:"
    for which in (0, 256, 128):
        O.gemm_force_kernel(which)
        tf = timeit(lambda: O.gemm_nt(x, w, out=y)); td = timeit(lambda: O.gemm_nt(dy, w, b_t=True, out=dx)); tw = timeit(lambda: O.gemm_nt(dy, x, a_t=True, b_t=True, out=dw))
        fl = 2.0 * R * N * K
        row += f"   [{which or 'auto':>4}] fwd {fl / tf / 1e9:5.0f} dgrad {fl / td / 1e9:5.0f} wgrad {fl / tw / 1e9:5.0f} TF ({(tf + td + tw) * 1e3:5.0f} us)"
        tot[which] += tf + td + tw
    print(row, flush=True)
O.gemm_force_kernel(0)
print("per tower layer (fwd + dgrad + wgrad, no epilogues): " + "  ".join(f"{k or 'auto'}: {v:.3f} ms" for k, v in tot.items()), " x 24 layers")
