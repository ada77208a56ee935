"""A/B of the fused dK|dV kernels at D = 128: attn_bwd2_kv_k<MODE 3> (rounds 2-4, LDS-DMA copies; switch value 1) against attn_bwd3_kv_k
(round 5: register-staged copies, three LDS stages, one barrier in the middle of a tile; switch value 2, the default), same process, interleaved
passes.  The two must agree BIT FOR BIT on dK and dV (same products, same per-accumulator summation order), on every geometry incl. ragged
lengths, sequence ends inside a tile and non-causal attention; then timed at cfg 3 / cfg 5 / ragged."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa: E402

from merlin_amd import _lib as L  # noqa: E402
from merlin_amd import ops as O  # noqa: E402


def mode(m):
    L.lib().mh_attn_bwd_fused_kv(C.c_int(m))


def make(B, S, H, D, seed, dtype=torch.bfloat16):
    g = torch.Generator(device="cuda").manual_seed(seed)
    qkv = torch.randn(B * S, 3 * H * D, generator=g, device="cuda").to(dtype)
    do = torch.randn(B * S, H * D, generator=g, device="cuda").to(dtype)
    return qkv, do


def bwd(qkv, do, B, S, H, D, causal, sl, rope=None, spill=False):
    q, k, v = (qkv[:, i * H * D:(i + 1) * H * D] for i in range(3))
    o, lse = O.attn_fwd2(q, k, v, B, S, H, D, causal, seqlens=sl)
    dqkv = torch.full_like(qkv, float("nan"))
    O.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, causal, seqlens=sl, dq=dqkv[:, :H * D], dk=dqkv[:, H * D:2 * H * D], dv=dqkv[:, 2 * H * D:], rope=rope,
                spill=spill)
    return dqkv


def ref_dq(qkv, do, B, S, H, D):
    """fp64 reference of dQ (causal softmax attention backward) for one small case."""
    q, k, v = (qkv[:, i * H * D:(i + 1) * H * D].double().view(B, S, H, D).transpose(1, 2) for i in range(3))
    q.requires_grad_(True)
    s = (q @ k.transpose(-1, -2)) / D ** 0.5
    s = s.masked_fill(~torch.ones(S, S, dtype=torch.bool, device=s.device).tril(), float("-inf"))
    o = torch.softmax(s, -1) @ v
    o.backward(do.double().view(B, S, H, D).transpose(1, 2))
    return q.grad.transpose(1, 2).reshape(B * S, H * D)


def check_spill():
    """five-product form (dS spilled by the dK|dV kernel) against the seven-product form: dK, dV bit-identical, dQ to rounding - and both
    dQ against an fp64 reference (the spill form must not be further away)."""
    bad = 0
    H, D = 4, 128
    mode(2)
    for dtype in (torch.bfloat16, torch.float16):
        for (B, S) in ((2, 128), (1, 256), (2, 384), (1, 1024), (3, 640), (1, 2048)):
            qkv, do = make(B, S, H, D, seed=7 * S + B, dtype=dtype)
            a7 = bwd(qkv, do, B, S, H, D, True, None, spill=False)
            a5 = bwd(qkv, do, B, S, H, D, True, None, spill=True)
            a5b = bwd(qkv, do, B, S, H, D, True, None, spill=True)
            torch.cuda.synchronize()
            kv_same = torch.equal(a7[:, H * D:], a5[:, H * D:])
            det = torch.equal(a5, a5b)
            ref = ref_dq(qkv, do, B, S, H, D)
            e7 = float((a7[:, :H * D].double() - ref).abs().max() / ref.abs().max())
            e5 = float((a5[:, :H * D].double() - ref).abs().max() / ref.abs().max())
            ok = kv_same and det and bool(torch.isfinite(a5.float()).all()) and e5 <= 1.1 * e7 + 1e-4
            if not ok:
                bad += 1
            print(f"spill check {str(dtype):15s} B={B} S={S}: dK|dV identical {kv_same}, deterministic {det}, dQ err vs fp64: 7-product {e7:.3e}  5-product {e5:.3e}  {'OK' if ok else 'BAD'}")
    return bad


def check():
    bad = 0
    H, D = 4, 128
    cases = [(2, 128, True, None), (2, 256, True, None), (1, 1024, True, None), (2, 1000, True, None), (2, 613, True, None), (3, 70, True, None),
             (2, 4096 // 8, False, None), (1, 777, False, None), (4, 512, True, [512, 300, 1, 129]), (3, 640, True, [640, 64, 577]),
             (2, 448, False, [448, 100]), (1, 192, True, [190]), (2, 2048, True, None), (2, 1088, True, [1088, 1025])]
    for dtype in (torch.bfloat16, torch.float16):
        for (B, S, causal, lens) in cases:
            qkv, do = make(B, S, H, D, seed=S + B, dtype=dtype)
            sl = torch.tensor(lens, dtype=torch.int32, device="cuda") if lens else None
            outs = []
            for m in (1, 2, 1, 2):
                mode(m)
                outs.append(bwd(qkv, do, B, S, H, D, causal, sl))
            torch.cuda.synchronize()
            same_run = torch.equal(outs[1], outs[3])  # deterministic
            eq = torch.equal(outs[0], outs[1])
            fin = bool(torch.isfinite(outs[1].float()).all())
            if not (eq and fin and same_run):
                bad += 1
                d = (outs[0].float() - outs[1].float()).abs()
                print(f"MISMATCH {dtype} B={B} S={S} causal={causal} lens={lens}: equal={eq} finite={fin} deterministic={same_run} max|d|={float(d.nan_to_num(1e9).max()):.3e} "
                      f"dq {float(d[:, :H * D].nan_to_num(1e9).max()):.2e} dk {float(d[:, H * D:2 * H * D].nan_to_num(1e9).max()):.2e} dv {float(d[:, 2 * H * D:].nan_to_num(1e9).max()):.2e}")
    print("bitwise check:", "OK" if not bad else f"{bad} mismatching cases")
    return bad


def timeit(fn, iters=20, warm=4):
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


def bench():
    H, D = 32, 128
    for tag, B, S, lens in (("cfg3", 8, 4096, None), ("cfg5", 4, 8192, None), ("ragged", 8, 4096, [4096, 3000, 4001, 65, 2048, 4095, 1, 3333])):
        qkv, do = make(B, S, H, D, seed=1)
        q, k, v = (qkv[:, i * H * D:(i + 1) * H * D] for i in range(3))
        sl = torch.tensor(lens, dtype=torch.int32, device="cuda") if lens else None
        o, lse = O.attn_fwd2(q, k, v, B, S, H, D, True, seqlens=sl)
        dq, dk, dv = (torch.empty_like(o) for _ in range(3))
        res = {1: [], 2: [], 3: []}
        for _ in range(3):
            for m in (1, 2, 3):
                mode(min(m, 2))
                res[m].append(timeit(lambda: O.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, True, seqlens=sl, dq=dq, dk=dk, dv=dv, spill=(m == 3))))
        print(f"{tag:7s} backward (delta + dK|dV + dQ) ms: LDS-DMA form " + " ".join(f"{t:.4f}" for t in res[1]) + "   register-staged form " +
              " ".join(f"{t:.4f}" for t in res[2]) + "   + dS spill (5 products) " + " ".join(f"{t:.4f}" for t in res[3]), flush=True)


if __name__ == "__main__":
    O.attn_bwd_spill(False)
    bad = check()
    bad += check_spill()
    bench()
    mode(2)
    sys.exit(1 if bad else 0)
