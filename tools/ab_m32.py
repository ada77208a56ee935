"""A/B of the GEMM main loop on 32x32x16 MFMAs (development arms, NT layout only): 256 = product kernel (16x16x32, four quadrant
phases per K-tile), 32 = 32x32x16 in the same quadrant phases, 33 = 32x32x16 with two phases per K-tile (four accumulators per phase).
MH_LIB_PATH=tools/dev_arms/libmerlin_hip_dev.so python tools/ab_m32.py   (run on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlin_amd import ops as O

dev = torch.device("cuda:0")
O.gemm_persistent(False)  # the arms are one block per tile: compare like with like


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


a = torch.randn(1000, 512, device=dev).bfloat16(); b = torch.randn(515, 512, device=dev).bfloat16()
O.gemm_force_kernel(256); r0 = O.gemm_nt(a, b, out_f32=True)
for k in (32, 33):
    O.gemm_force_kernel(k); r = O.gemm_nt(a, b, out_f32=True)
    print(f"kernel {k} vs 256: max rel diff {float((r - r0).abs().max() / r0.abs().max()):.2e}", flush=True)
T = 32768
for (M, N, K) in ((T, 12288, 4096), (T, 4096, 4096), (T, 22016, 4096), (T, 4096, 11008), (8192, 8192, 8192)):
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    row = f"[{M},{N},{K}]"
    for rep in range(2):
        for k in (256, 32, 33):
            O.gemm_force_kernel(k)
            t = timeit(lambda: O.gemm_nt(A, B, out=C))
            row += f"  k{k}: {2.0 * M * N * K / t / 1e9:.0f}"
    print(row + "  TFLOP/s", flush=True)
O.gemm_force_kernel(0)
