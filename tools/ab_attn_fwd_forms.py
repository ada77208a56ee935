"""D = 128 attention forward, the three forms of the product library side by side (mh_attn_fwd_pingpong 0 / 1 / 2 = attn_fwd2 / attn_fwd3 /
attn_fwd4) at the benchmark geometries: ms per call, TFLOP/s of the causal / full product, and the outputs' checksums (forms must agree
bit for bit with form 0).  Two interleaved passes."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa: E402

from merlin_amd import ops as O  # noqa: E402

D = 128


def timeit(fn, iters=30, warm=8):
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


CASES = [("cfg3", 8, 4096, 32, True, None), ("cfg5", 4, 8192, 32, True, None),
         ("ragged", 8, 4096, 32, True, [4096, 3000, 4001, 65, 2048, 4095, 1, 3333]), ("full-4k", 8, 4096, 32, False, None), ("cfg2", 1, 613, 32, True, None)]

if __name__ == "__main__":
    data = {}
    for tag, B, S, H, causal, lens in CASES:
        g = torch.Generator(device="cuda").manual_seed(S + B)
        qkv = torch.randn(B * S, 3 * H * D, generator=g, device="cuda").to(torch.bfloat16)
        sl = torch.tensor(lens, dtype=torch.int32, device="cuda") if lens is not None else None
        data[tag] = (qkv, sl)
    for rep in range(2):
        for tag, B, S, H, causal, lens in CASES:
            qkv, sl = data[tag]
            q, k, v = (qkv[:, i * H * D:(i + 1) * H * D] for i in range(3))
            pairs = sum(min(l, S) * (min(l, S) + 1) // 2 if causal else min(l, S) ** 2 for l in (lens or [S] * B))
            flop = 4.0 * D * H * pairs
            ref = None
            line = f"{tag:8s} B={B} S={S} causal={int(causal)}:"
            for form in (0, 1, 2):
                O.attn_fwd_pingpong(form)
                o, lse = O.attn_fwd2(q, k, v, B, S, H, D, causal, seqlens=sl)
                t = timeit(lambda: O.attn_fwd2(q, k, v, B, S, H, D, causal, seqlens=sl, out=o, lse=lse))
                same = "ref" if ref is None else ("bit-identical" if torch.equal(o, ref) else f"DIFFERS {float((o.float() - ref.float()).abs().max()):.2e}")
                if ref is None:
                    ref = o.clone()
                line += f"  form {form}: {t:.4f} ms {flop / t / 1e9:7.1f} TF ({same})"
            O.attn_fwd_pingpong(0)
            print(line, flush=True)
