"""Where the fused dK|dV kernel's wave cycles go, segment by segment (development probe).

tools/probes/attn_bwd2_timing.so = csrc/attn_bwd2.hip compiled with -DMH_KV_TIMING: the kernel stamps `s_memtime` at the seams of a
full tile (tile top -> after vmcnt(0) + barrier -> after the next tile's copy requests and the fragment addresses -> after segments
A, B, C; segment D is closed by the next tile's first stamp) and every wave adds the differences up.  The instrumentation itself costs
time (MI355X_MICROARCH: ~11 %); the SPLIT is what this is for.  Printed per block class (causal blocks differ 32 : 1 in tile count).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMH_KV_TIMING -I merlin_amd/csrc -shared merlin_amd/csrc/attn_bwd2.hip -o tools/probes/attn_bwd2_timing.so
    python tools/probes/kv_timing.py [B S]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa: E402

from merlin_amd import ops as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    B, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 4096)
    H, D = 32, 128
    lib = C.CDLL(os.path.join(HERE, "attn_bwd2_timing.so"))
    form = int(os.environ.get("KV_FORM", "2"))  # 1: attn_bwd2_kv_k<MODE 3> (LDS-DMA copies), 2: attn_bwd3_kv_k (register-staged)
    lib.mh_attn_bwd_fused_kv(C.c_int(form))
    print("kernel form:", form)
    g = torch.Generator(device="cuda").manual_seed(S + B)
    qkv = torch.randn(B * S, 3 * H * D, generator=g, device="cuda").to(torch.bfloat16)
    q, k, v = (qkv[:, i * H * D:(i + 1) * H * D] for i in range(3))
    do = torch.randn(B * S, H * D, generator=g, device="cuda").to(torch.bfloat16)
    o, lse = O.attn_fwd2(q, k, v, B, S, H, D, True)
    dq, dk, dv = (torch.empty(B * S, H * D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    delta = torch.zeros(2, B, H, S, dtype=torch.float32, device="cuda")
    nblk = B * H * (S // 128)
    dbg = torch.zeros(nblk * 4 * 8, dtype=torch.int64, device="cuda")
    lib.mh_kv_timing_buffer(C.c_void_p(dbg.data_ptr()))
    p = lambda t: C.c_void_p(0 if t is None else t.data_ptr())  # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run():
        rc = lib.mh_attn_bwd2(p(q), C.c_int64(q.stride(0)), p(k), C.c_int64(k.stride(0)), p(v), C.c_int64(v.stride(0)), p(o), C.c_int64(o.stride(0)),
                              p(do), C.c_int64(do.stride(0)), p(lse), p(delta), p(dq), C.c_int64(dq.stride(0)), p(dk), C.c_int64(dk.stride(0)),
                              p(dv), C.c_int64(dv.stride(0)), p(None), C.c_int(B), C.c_int(S), C.c_int(H), C.c_int(D), C.c_int(1), p(None), C.c_int(0), st)
        assert rc == 0, rc

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t = dbg.view(nblk, 4, 8).double().cpu()
    # product library on the same inputs: the instrumented build must agree bit for bit
    dq2, dk2, dv2 = O.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, True)
    print("bit-identical to the product kernels:", bool(torch.equal(dk, dk2) and torch.equal(dv, dv2) and torch.equal(dq, dq2)))
    names = ["D (closing seg.)", "vmcnt(0)+barrier" if form == 1 else "barrier (mid-tile)", "copies+addresses" if form == 1 else "-", "A: S,dP h0",
             "B: S,dP h1 | elem h0", "C: dV,dK h0 | elem h1"]
    ntile = t[:, :, 6]
    for lo, hi in ((57, 65), (27, 34), (6, 12), (1, 4)):
        sel = (ntile[:, 0] >= lo) & (ntile[:, 0] < hi)
        if not bool(sel.any()):
            continue
        tt = t[sel]
        n = tt[:, :, 6].mean()
        per = [float((tt[:, :, i] / tt[:, :, 6]).mean()) for i in range(6)]
        tot = sum(per)
        print(f"blocks with {lo}..{hi - 1} full tiles ({int(sel.sum())} blocks, mean {float(n):.1f} tiles): {tot:.0f} cycles per full tile (64 MFMAs = 2048 at one per 32 cycles)")
        for nm, c in zip(names, per):
            print(f"    {nm:24s} {c:8.1f} cycles  {100 * c / tot:5.1f} %")
        w = [float((tt[:, wv, :6].sum(-1) / tt[:, wv, 6]).mean()) for wv in range(4)]
        print("    per wave:", " ".join(f"{x:.0f}" for x in w))


if __name__ == "__main__":
    main()
