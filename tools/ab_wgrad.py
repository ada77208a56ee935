import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlin_amd import ops as O, _lib as L
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_ops import timeit
dev = torch.device("cuda:0")
T = 27696
for (No, Ki) in [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096), (4096, 4096)]:
    dy = torch.randn(T, No, device=dev).bfloat16(); x = torch.randn(T, Ki, device=dev).bfloat16()
    out = torch.empty(No, Ki, device=dev, dtype=torch.bfloat16)
    s = int(L.lib().mh_gemm_splitk_max(No, Ki, T))
    t = timeit(lambda: O.wgrad_tn(dy, x, out, accum=False))
    print(f"wgrad T={T} No={No} Ki={Ki}: splits {s}  {t*1e3:.3f} ms  {2.0*T*No*Ki/t/1e12:.0f} TF", flush=True)
    for sp in (1, 2, 4, 8):
        ws = torch.empty(sp * No * Ki, device=dev)
        def f():
            L.check(L.lib().mh_gemm_splitk(O.p(dy), O.i64(No), O.i32(1), O.p(x), O.i64(Ki), O.i32(1), O.p(out), O.i64(Ki), O.i32(No), O.i32(Ki), O.i32(T),
                                           O.i32(O.dt_of(dy)), O.i32(0), O.i32(0), O.i32(sp), O.p(ws), O._stream()), "x")
        t = timeit(f)
        print(f"     splits={sp}: {t*1e3:.3f} ms {2.0*T*No*Ki/t/1e12:.0f} TF", flush=True)
