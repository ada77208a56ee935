"""Probe (MI355X only): operand layout, scale semantics and issue rate of v_mfma_scale_f32_16x16x128_f8f6f4 (the gfx950
block-scaled fp8 MFMA, 2x the bf16 rate on paper) - groundwork for the fp8 weight path of BASELINE cfg 5.
Hypothesis tested: lane l holds row (l & 15) and the 32 consecutive k = 32*(l >> 4) .. +31 of A (8 VGPRs, little-endian
bytes), same for B (columns); D lane l reg r = D[4*(l>>4) + r][l & 15] like the 16x16x32 MFMAs; the scale operands are
E8M0 exponents (value 2^(s-127)) read from byte `opsel` of a per-lane VGPR and apply to that lane's 32-element block.
usage: python tools/probes/f8f6f4_probe.py      (compiles tools/probes/f8f6f4_probe.hip with hipcc, loads it via ctypes)"""
import ctypes as C
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
src, so = os.path.join(HERE, "f8f6f4_probe.hip"), os.path.join(HERE, "f8f6f4_probe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", so])
lib = C.CDLL(so)
dev = "cuda"
torch.manual_seed(0)
A = (torch.randn(16, 128, device=dev) * 2).to(torch.float8_e4m3fn)
B = (torch.randn(16, 128, device=dev) * 2).to(torch.float8_e4m3fn)
# per-lane operand images under the hypothesis: lane l -> row l&15, k-block l>>4
lanes = torch.arange(64, device=dev)
a_img = A.view(torch.uint8)[lanes & 15].view(64, 4, 32)[lanes, lanes >> 4].contiguous()   # [64, 32] bytes
b_img = B.view(torch.uint8)[lanes & 15].view(64, 4, 32)[lanes, lanes >> 4].contiguous()
out = torch.zeros(64, 4, device=dev)
sa = torch.full((64,), 127, dtype=torch.int32, device=dev)
sb = torch.full((64,), 127, dtype=torch.int32, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
lib.run_once(p(a_img), p(b_img), p(sa), p(sb), p(out))
torch.cuda.synchronize()
ref = A.float() @ B.float().t()                      # [16 rows of A, 16 rows of B]
got = torch.zeros(16, 16, device=dev)
for l in range(64):
    for r in range(4):
        got[4 * (l >> 4) + r, l & 15] = out[l, r]
e1 = float((got - ref).abs().max() / ref.abs().max())
e2 = float((got - ref.t()).abs().max() / ref.abs().max())
print(f"layout hypothesis: D = A B^T rel err {e1:.2e}   (transposed reading: {e2:.2e})")
# scale semantics: bump the A scale exponent (+1 = x2) of ONE lane at a time and find which (row, 32-k block) doubled
def unpack(o):
    g = torch.zeros(16, 16, device=dev)
    for l in range(64):
        for r in range(4):
            g[4 * (l >> 4) + r, l & 15] = o[l, r]
    return g


base = unpack(out)
contrib = torch.stack([A.float()[:, 32 * j:32 * j + 32] @ B.float()[:, 32 * j:32 * j + 32].t() for j in range(4)])  # [4, 16, 16]
mapping = {}
for L in range(64):
    sa2 = sa.clone(); sa2[L] = 128
    o2 = torch.zeros_like(out)
    lib.run_once(p(a_img), p(b_img), p(sa2), p(sb), p(o2))
    torch.cuda.synchronize()
    d = unpack(o2) - base
    rows = [r for r in range(16) if float(d[r].abs().max()) > 1e-6]
    hit = None
    for r in rows:
        for j in range(4):
            if float((d[r] - contrib[j, r]).abs().max()) < 1e-3 * float(contrib[j, r].abs().max() + 1e-6):
                hit = (r, j)
    mapping[L] = (rows, hit)
# exact k set scaled by each lane's scale: B_t[n, k] = 1 at k = 16 t + n  ->  D[r, n] = A[r, 16 t + n] * scale(r, k)
def img(M_):
    return M_.view(torch.uint8)[lanes & 15].view(64, 4, 32)[lanes, lanes >> 4].contiguous()


ksets = {}
for L in (0, 16, 32, 48, 5, 21):
    sa2 = sa.clone(); sa2[L] = 128
    doubled = []
    for t in range(8):
        Bt = torch.zeros(16, 128, device=dev)
        for n_ in range(16):
            Bt[n_, 16 * t + n_] = 1.0
        Bt8 = Bt.to(torch.float8_e4m3fn)
        o1, o2 = torch.zeros_like(out), torch.zeros_like(out)
        lib.run_once(p(a_img), p(img(Bt8)), p(sa), p(sb), p(o1))
        lib.run_once(p(a_img), p(img(Bt8)), p(sa2), p(sb), p(o2))
        torch.cuda.synchronize()
        g1, g2 = unpack(o1), unpack(o2)
        r = L & 15
        for n_ in range(16):
            if float(g1[r, n_]) != 0 and abs(float(g2[r, n_]) / float(g1[r, n_]) - 2.0) < 1e-3:
                doubled.append(16 * t + n_)
    ksets[L] = doubled
    print(f"A-scale of lane {L:2d} (row {L & 15}) doubles k in {doubled[:4]}..{doubled[-4:] if doubled else []}  ({len(doubled)} values)")
# issue rate against the bf16 16x16x32 MFMA
n = 1 << 16
t = torch.zeros(2, device=dev)
for which in (0, 1):
    for _ in range(2):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); lib.run_rate(C.c_int(which), C.c_int(n)); e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e)
    flops = (2 * 16 * 16 * (128 if which else 32)) * 16.0 * n * 4 * 256  # 16 MFMAs per loop iteration, 4 waves x 256 blocks... see .hip
    print(("f8f6f4 16x16x128" if which else "bf16  16x16x32 "), f"{ms:.3f} ms  -> {flops / ms / 1e9:.0f} TFLOP/s aggregate (256 CUs x 4 waves)")
