"""GEMV bandwidth at the decode step's shapes: one-wave-per-row kernel vs the MFMA form, 16-bit and fp8 weights (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlin_amd import ops as O

dev = torch.device("cuda:0")
if os.environ.get("MH_GEMV_MFMA_WIDE"):  # A/B: 0 = 8 waves per block in the small-N MFMA form
    O.gemv_mfma_wide(os.environ["MH_GEMV_MFMA_WIDE"] != "0")
MS = tuple(int(a) for a in sys.argv[1:]) or (1, 2, 4, 8, 16)


def timeit(fn, iters=20, warm=5):
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


for (N, K) in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008), (32064, 4096)):
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    q8 = O.quant_fp8_b128(w)
    # rotate over several weight copies so the 256 MB Infinity Cache does not serve the stream
    ws = [w.clone() for _ in range(max(1, int(600e6 // (N * K * 2))))]
    q8s = [(q8[0].clone(), q8[1].clone()) for _ in range(max(1, int(600e6 // (N * K))))]
    for M in MS:
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        row = f"N={N:5d} K={K:5d} M={M:2d}:"
        for mode, mn in (("row-wave", 17), ("mfma", 3)):
            if M > 8 and mn == 17:
                row += "   row-wave  n/a          "
                continue
            if M < 3 and mn == 3:
                row += "   mfma      n/a          "
                continue
            O.gemv_mfma_min_rows(mn)
            it = [0]
            def f():
                it[0] += 1
                O.gemv(x, ws[it[0] % len(ws)])
            t = timeit(f)
            it = [0]
            def g():
                it[0] += 1
                O.gemv_fp8w(x, q8s[it[0] % len(q8s)])
            t8 = timeit(g)
            row += f"   {mode:8s} {N * K * 2 / t / 1e9:5.2f} TB/s | fp8 {N * K / t8 / 1e9:5.2f} TB/s ({t8 * 1e3:5.1f} us)"
        print(row, flush=True)
O.gemv_mfma_min_rows(3)
