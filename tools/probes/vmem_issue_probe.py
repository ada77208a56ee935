"""ns per 'copy slot' (PER MFMAs + one vector-memory instruction) of tools/probes/vmem_issue_probe.hip, per form of the instruction, at one and two waves per SIMD.
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/vmem_issue_probe.hip -o tools/probes/vmem_issue_probe.so ; python tools/probes/vmem_issue_probe.py"""
import ctypes
import os

import torch

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "vmem_issue_probe.so"))
lib.vmem_issue_probe.restype = ctypes.c_float
dev = torch.device("cuda:0")
src = (torch.randn(300 << 18, device=dev)).contiguous()  # 300 MiB of fp32
sink = torch.zeros(256 * 512, device=dev)
names = {0: "MFMAs only", 1: "m0 + buffer_load_dwordx4 offen lds (the GEMM's copy)", 2: "buffer_load_dwordx4 offen -> registers", 3: "buffer_load_dwordx2 offen -> registers",
         4: "buffer_load_dword offen -> registers", 5: "m0 + global_load_lds_dwordx4 (saddr)", 6: "the GEMM's copy, 32 active lanes", 7: "buffer_load_dwordx4 off (wave-uniform address)",
         8: "the GEMM's copy in bursts of two (same count)", 9: "ds_read_b128 (a fragment read, for scale)"}


WINDOW = 64 << 10


def t(mode, per, waves, blocks=256, iters=2000, window=None):
    return lib.vmem_issue_probe(mode, per, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(sink.data_ptr()), iters, waves, blocks, 5, ctypes.c_uint(window or WINDOW))


for waves in (4, 8):
    for per in (8, 4, 16):
        base = t(0, per, waves)
        print(f"--- {waves} waves per CU ({waves // 4} per SIMD), {per} MFMAs per slot: MFMAs only {base:7.2f} ns per slot = {base / per:5.2f} ns per MFMA", flush=True)
        for mode in range(1, 10):
            v = t(mode, per, waves)
            print(f"    {names[mode]:58s} {v:7.2f} ns per slot  (+{v - base:6.2f} ns per instruction)", flush=True)
print("--- GEMM-like surroundings (4 waves, 8 MFMAs per slot, L2-resident): +2 ds_read_b128 per slot and lgkmcnt(0) every 8 slots (G1), + s_barrier there (G2), + vmcnt(8) in front of it (G3)")
for gl in (1, 2, 3):
    b = t(100 * gl, 8, 4)
    v1, v2 = t(100 * gl + 1, 8, 4), t(100 * gl + 2, 8, 4)
    print(f"    G{gl}: no copy {b:7.2f} ns per slot | the GEMM's copy {v1:7.2f} (+{v1 - b:6.2f}) | into registers {v2:7.2f} (+{v2 - b:6.2f})", flush=True)
    b = t(100 * gl, 8, 4, window=1 << 20)
    v1 = t(100 * gl + 1, 8, 4, window=1 << 20)
    print(f"        1 MiB per CU:                          the GEMM's copy {v1:7.2f} (+{v1 - b:6.2f})", flush=True)
b = t(100, 8, 4)
for mode, nm in ((1, "the GEMM's copy"), (2, "dwordx4 into registers"), (4, "dword into registers"), (6, "the copy, 32 active lanes"), (11, "the copy as two 32-lane halves"),
                 (7, "dwordx4 into registers, wave-uniform address (off)"), (10, "LDS-DMA dwordx4, wave-uniform address (off)"), (8, "the copy in bursts of two")):
    v = t(100 + mode, 8, 4)
    print(f"    G1 + {nm:52s} {v:7.2f} ns per slot (+{v - b:6.2f})", flush=True)
print("--- footprint: 1 MiB per CU (256 MiB for the chip: Infinity Cache / HBM instead of the L2s), 4 waves, 8 MFMAs per slot")
b = t(0, 8, 4)
for mode in (1, 2, 4, 6):
    v = t(mode, 8, 4, window=1 << 20)
    print(f"    {names[mode]:58s} {v:7.2f} ns per slot  (+{v - b:6.2f})")
print("--- 32 CUs only (4 waves, 8 MFMAs per slot)")
b = t(0, 8, 4, blocks=32)
for mode in (1, 2):
    v = t(mode, 8, 4, blocks=32)
    print(f"    {names[mode]:58s} {v:7.2f} ns per slot  (+{v - b:6.2f})")
