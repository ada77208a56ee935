"""Where a tile of the 4-wave GEMM spends its time, block by block (development probe).

tools/probes/libmerlin_w4_timing.so = the library compiled with -DMH_W4_TIMING: every gemm_w4 block stamps `s_memrealtime` (100 MHz, one
clock for the whole chip) at entry, when tile 0's operands have landed and its first fragments are in registers, at the end of the main loop and
after the store phase, plus HW_ID / XCC_ID.  From the stamps: prologue, main loop, store phase per block, and - per CU - the gap between the end
of one block and the entry of the next one dispatched there.

    python tools/probes/w4_timing.py build        # here (cross-compiles)
    MH_LIB_PATH=tools/probes/libmerlin_w4_timing.so python tools/probes/w4_timing.py     # on the GPU box
"""
import ctypes as C
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
LIBT = os.path.join(HERE, "libmerlin_w4_timing.so")

if len(sys.argv) > 1 and sys.argv[1] == "build":
    import concurrent.futures as cf
    import subprocess

    from merlin_amd.csrc import build as B

    objdir = os.path.join(HERE, "build_w4t")
    with cf.ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(lambda s: B._compile(s, False, objdir=objdir, extra=("-DMH_W4_TIMING",)), B.SOURCES))
    subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIBT], check=True)
    print("built", LIBT)
    sys.exit(0)

import torch  # noqa: E402

from merlin_amd import _lib as L  # noqa: E402
from merlin_amd import ops as O  # noqa: E402


def main():
    lib = L.lib()
    assert hasattr(lib, "mh_w4_timing_buffer"), "run with MH_LIB_PATH=tools/probes/libmerlin_w4_timing.so"
    T, d, ff = 32768, 4096, 11008
    dev = torch.device("cuda:0")
    rnd = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)  # noqa: E731
    x, xf = rnd(T, d), rnd(T, ff)
    wo, wqkv, wd = rnd(d, d, scale=0.02), rnd(3 * d, d, scale=0.02), rnd(d, ff, scale=0.02)
    dy = rnd(T, d)
    resid = rnd(T, d)
    wgu = rnd(2 * ff, d, scale=0.02)
    rope = O.rope_table(4096, 128, 10000.0, dev)
    x32 = torch.zeros(T, d, dtype=torch.float32, device=dev)
    nt = T // 256
    cases = [("NN dgrad o (16-bit store) [T,4096,4096]", lambda: O.gemm_nt(x, wo, b_t=True), nt * (d // 256)),
             ("NT o-proj into the fp32 stream [T,4096,4096]", lambda: O.gemm_nt(x, wo, out=x32, accum=True), nt * (d // 256)),
             ("NT q|k|v + RoPE [T,12288,4096]", lambda: O.gemm_nt_rope(x, wqkv, rope, 4096, 32, 128), nt * (3 * d // 256)),
             ("NT gate|up + SwiGLU [T,22016,4096]", lambda: O.gemm_swiglu_fwd(x, wgu), nt * (ff // 128)),
             ("NT down into the fp32 stream [T,4096,11008]", lambda: O.gemm_nt(xf, wd, out=x32, accum=True), nt * (d // 256)),
             ("NT lm_head-like fp32 store [T,4096,4096]", lambda: O.gemm_nt(x, wo, out_f32=True), nt * (d // 256)),
             ("TN wgrad o [4096,4096,T]", lambda: O.wgrad_tn(dy, x, torch.empty(d, d, dtype=torch.bfloat16, device=dev), accum=False), (d // 256) * (d // 256))]
    for name, fn, nblk in cases:
        for _ in range(3):
            fn()
        dbg = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
        assert lib.mh_w4_timing_buffer(C.c_void_p(dbg.data_ptr())) == 0
        torch.cuda.synchronize()
        fn()
        torch.cuda.synchronize()
        lib.mh_w4_timing_buffer(C.c_void_p(0))
        t = dbg.view(nblk, 8).cpu()
        if int((t[:, 0] == 0).sum()):
            print(name, ": blocks without stamps:", int((t[:, 0] == 0).sum()))
            continue
        t0 = int(t[:, 0].min())
        ent, lan, mainl, end = ((t[:, k] - t0).double() / 100.0 for k in range(4))  # microseconds
        cu = ((t[:, 4] >> 32) & 0xf) * 65536 + (t[:, 4] & 0xff00)  # XCC_ID | SE / SH / CU bits of HW_ID
        print(f"{name}: {nblk} blocks, kernel span {float(end.max()):.1f} us")
        print(f"   prologue (entry -> first fragments)  mean {float((lan - ent).mean()):6.2f} us   p10 {float((lan - ent).quantile(0.1)):6.2f}  p90 {float((lan - ent).quantile(0.9)):6.2f}")
        print(f"   main loop                            mean {float((mainl - lan).mean()):6.2f} us   p10 {float((mainl - lan).quantile(0.1)):6.2f}  p90 {float((mainl - lan).quantile(0.9)):6.2f}")
        print(f"   store phase                          mean {float((end - mainl).mean()):6.2f} us   p10 {float((end - mainl).quantile(0.1)):6.2f}  p90 {float((end - mainl).quantile(0.9)):6.2f}")
        if int((t[:, 5] != 0).sum()) == nblk and int((t[:, 6] != 0).sum()) == nblk:  # staged 16-bit store: barrier | pack + LDS writes | LDS reads + global stores
            b5, b6 = (t[:, 5] - t0).double() / 100.0, (t[:, 6] - t0).double() / 100.0
            print(f"      of it: barrier {float((b5 - mainl).mean()):5.2f}   pack + stage {float((b6 - b5).mean()):5.2f}   read back + store {float((end - b6).mean()):5.2f}")
        # per CU: order its blocks by entry; gap = next entry - previous end
        gaps, per_cu = [], {}
        for i in range(nblk):
            per_cu.setdefault(int(cu[i]), []).append((float(ent[i]), float(end[i])))
        for v in per_cu.values():
            v.sort()
            gaps += [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
        if gaps:
            g = torch.tensor(gaps)
            print(f"   {len(per_cu)} distinct CUs; gap end -> next entry on the same one: mean {float(g.mean()):6.2f} us  p10 {float(g.quantile(0.1)):6.2f}  p90 {float(g.quantile(0.9)):6.2f}")
        print(f"   first-round entries spread over {float(ent[:256].max() - ent[:256].min()):.2f} us; sum of means per block {float((end - ent).mean()):.2f} us", flush=True)


if __name__ == "__main__":
    main()
