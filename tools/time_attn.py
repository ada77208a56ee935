"""Attention forward / backward at the benchmark geometries with whichever library MH_LIB_PATH selects (default: the product library):
cfg 3 (B = 8, S = 4096, causal), cfg 5 (B = 4, S = 8192, causal), a ragged packed batch, the CLIP tower (48 x 577, D = 64, no mask).
Prints ms per call and a checksum of the outputs (A/B arms must agree bit for bit)."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa: E402

from merlin_amd import ops as O  # noqa: E402


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


def run(B, S, H, D, causal, seqlens=None, tag=""):
    g = torch.Generator(device="cuda").manual_seed(S + B)
    qkv = torch.randn(B * S, 3 * H * D, generator=g, device="cuda").to(torch.bfloat16)
    q, k, v = (qkv[:, i * H * D:(i + 1) * H * D] for i in range(3))
    do = torch.randn(B * S, H * D, generator=g, device="cuda").to(torch.bfloat16)
    sl = torch.tensor(seqlens, dtype=torch.int32, device="cuda") if seqlens is not None else None
    o, lse = O.attn_fwd2(q, k, v, B, S, H, D, causal, seqlens=sl)
    dq, dk, dv = O.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, causal, seqlens=sl)
    cs = [float(t.float().abs().sum()) for t in (o, dq, dk, dv)]
    tf = timeit(lambda: O.attn_fwd2(q, k, v, B, S, H, D, causal, seqlens=sl, out=o, lse=lse))
    tb = timeit(lambda: O.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, causal, seqlens=sl, dq=dq, dk=dk, dv=dv), iters=10, warm=3)
    print(f"{tag:14s} B={B} S={S} H={H} D={D} causal={int(causal)}: fwd {tf:.4f} ms  bwd {tb:.4f} ms  sums " + " ".join(f"{c:.6e}" for c in cs), flush=True)


if __name__ == "__main__":
    print("library:", os.environ.get("MH_LIB_PATH", "product"))
    form = int(os.environ.get("MH_ATTN_FWD_FORM", "0"))  # 0: attn_fwd2 (default), 1: attn_fwd3 (ping-pong), 2: attn_fwd4 (one wave per SIMD)
    if form:
        O.attn_fwd_pingpong(form)
        print("forward form:", form)
    for _ in range(2):
        run(8, 4096, 32, 128, True, tag="cfg3")
        run(4, 8192, 32, 128, True, tag="cfg5")
        run(8, 4096, 32, 128, True, seqlens=[4096, 3000, 4001, 65, 2048, 4095, 1, 3333], tag="ragged")
        run(48, 577, 16, 64, False, tag="tower")
