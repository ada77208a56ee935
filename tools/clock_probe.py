"""Sustained shader clock / power while one kernel family runs in a loop (rocm-smi sampled from a side thread).
python tools/clock_probe.py attn|gemm|idle"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlin_amd import ops as O

what = sys.argv[1] if len(sys.argv) > 1 else "attn"
dev = torch.device("cuda:0")
B, S, H, D = 8, 4096, 32, 128
qkv = torch.randn(B * S, 3 * H * D, device=dev).bfloat16()
q, k, v = (qkv[:, i * H * D:(i + 1) * H * D] for i in range(3))
do = torch.randn(B * S, H * D, device=dev).bfloat16()
x = torch.randn(32768, 4096, device=dev).bfloat16()
w = (torch.randn(12288, 4096, device=dev) * 0.02).bfloat16()
o, lse = O.attn_fwd2(q, k, v, B, S, H, D, True)
dq, dk, dv = O.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, True)
y = O.gemm_nt(x, w)
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=10).stdout
            keep = [l.strip() for l in out.splitlines() if ("sclk" in l or "Power" in l or "junction" in l.lower())]
            samples.append(" | ".join(keep))
        except Exception as e:  # noqa
            samples.append(repr(e))
        time.sleep(0.3)


th = threading.Thread(target=sampler)
th.start()
t0 = time.time()
n = 0
while time.time() - t0 < 6.0:
    for _ in range(10):
        if what.startswith("attn"):
            O.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, True, dq=dq, dk=dk, dv=dv)
        elif what == "gemm":
            O.gemm_nt(x, w, out=y)
        n += 1
    torch.cuda.synchronize()
stop = True
th.join()
print(what, "iterations", n, "ms/iter", (time.time() - t0) / max(n, 1) * 1e3)
for s_ in samples[2:14]:
    print("  ", s_[:220])
