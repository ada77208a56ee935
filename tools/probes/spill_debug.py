"""Development check of the dS spill layout (csrc/attn_bwd2.hip, ds_unit): decode the scratch buffer the dK|dV kernel wrote and compare it with
dS = P o (dP - delta) computed in torch; then dQ = scale * dS K from the DECODED buffer against the kernel's dQ (isolates writer from reader)."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa: E402

from merlin_amd import ops as O  # noqa: E402


def main():
    B, S, H, D = 1, int(sys.argv[1]) if len(sys.argv) > 1 else 256, 2, 128
    g = torch.Generator(device="cuda").manual_seed(3)
    qkv = torch.randn(B * S, 3 * H * D, generator=g, device="cuda").to(torch.bfloat16)
    do = torch.randn(B * S, H * D, generator=g, device="cuda").to(torch.bfloat16)
    q, k, v = (qkv[:, i * H * D:(i + 1) * H * D] for i in range(3))
    o, lse = O.attn_fwd2(q, k, v, B, S, H, D, True)
    dq, dk, dv = O.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, True, spill=True)
    torch.cuda.synchronize()
    ws = next(iter(O._spill_cache.values()))
    nkb, nqt = S // 128, S // 64
    units = nkb * nkb + nkb
    raw = ws[: B * H * units * 16384].view(torch.bfloat16).view(B * H, units, 4, 2, 2, 64, 8).float()  # [bh][unit][wave][hf][ks][lane][8 q]
    # reference dS [bh, key, query]
    qf, kf, vf, dof, of = (t.float().view(B, S, H, D).transpose(1, 2).reshape(B * H, S, D) for t in (q, k, v, do, o))
    s = qf @ kf.transpose(1, 2) / D ** 0.5
    mask = torch.ones(S, S, dtype=torch.bool, device="cuda").tril()
    p = torch.exp(s - lse.view(B * H, -1)[:, :S, None]).masked_fill(~mask, 0.0)
    dp = dof @ vf.transpose(1, 2)
    delta = (dof * of).sum(-1, keepdim=True)
    ds_ref = (p * (dp - delta)).transpose(1, 2)  # [bh, key, query]
    dec = torch.zeros(B * H, S, S, device="cuda")
    for kb in range(nkb):
        for qt in range(2 * kb, nqt):
            u = kb * nqt - kb * (kb - 1) + (qt - 2 * kb)
            blk = raw[:, u]  # [bh][wave][hf][ks][lane][8]
            for w in range(4):
                for hf in range(2):
                    for ks in range(2):
                        for hi in range(2):
                            keys = 128 * kb + 32 * w + torch.arange(32, device="cuda")
                            q0 = 64 * qt + 32 * hf + 16 * ks + 8 * hi
                            dec[:, keys, q0:q0 + 8] = blk[:, w, hf, ks, hi * 32:(hi + 1) * 32, :]
    err = (dec - ds_ref).abs().max() / ds_ref.abs().max()
    print(f"S={S}: decoded spill buffer vs torch dS: max rel err {float(err):.3e}")
    bad = ((dec - ds_ref).abs() > 0.02 * ds_ref.abs().max()).nonzero()
    if bad.numel():
        print("first mismatches (bh, key, query):", bad[:10].tolist(), " count", bad.shape[0])
    # the same comparison in the buffer's own coordinates
    enc = torch.zeros_like(raw)
    for kb in range(nkb):
        for qt in range(2 * kb, nqt):
            u = kb * nqt - kb * (kb - 1) + (qt - 2 * kb)
            for w in range(4):
                for hf in range(2):
                    for ks in range(2):
                        for hi in range(2):
                            keys = 128 * kb + 32 * w + torch.arange(32, device="cuda")
                            q0 = 64 * qt + 32 * hf + 16 * ks + 8 * hi
                            enc[:, u, w, hf, ks, hi * 32:(hi + 1) * 32, :] = ds_ref[:, keys, q0:q0 + 8]
    badc = ((raw - enc).abs() > 0.02 * ds_ref.abs().max()).nonzero()
    print("mismatches in buffer coordinates (bh, unit, wave, hf, ks, lane, e):", badc[:24].tolist(), "count", badc.shape[0])
    import collections
    print("by (unit, wave, hf, ks):", sorted(collections.Counter((int(r[1]), int(r[2]), int(r[3]), int(r[4])) for r in badc.tolist()).items()))
    print("by lane:", sorted(collections.Counter(int(r[5]) for r in badc.tolist()).items()))
    dq_dec = (dec.transpose(1, 2) @ kf) / D ** 0.5  # [bh, q, d]
    dq_k = dq.float().view(B, S, H, D).transpose(1, 2).reshape(B * H, S, D)
    print(f"dQ from the decoded buffer vs kernel dQ: max rel err {float((dq_dec - dq_k).abs().max() / dq_dec.abs().max()):.3e}")
    e = (dq_dec - dq_k).abs()
    rows = (e.amax(-1) > 0.02 * dq_dec.abs().max()).nonzero()
    print("query rows off:", rows[:20].tolist(), "count", rows.shape[0])
    cols = (e.amax(1) > 0.02 * dq_dec.abs().max()).nonzero()
    print("(bh, d) columns off:", cols.shape[0])


if __name__ == "__main__":
    main()
