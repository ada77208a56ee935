"""Wrappers for the development-only A/B arms (first-generation attention kernels).  Needs the dev library:
    python -m merlin_amd.csrc.build --dev
    MH_LIB_PATH=tools/dev_arms/libmerlin_hip_dev.so python tools/...
Not part of the product package."""
import torch

from merlin_amd import _lib as L
from merlin_amd._lib import i32, i64, p
from merlin_amd.ops import _stream, dt_of, round_up


def attn_prep_v(v, B, S, H, D, out=None):
    """v: [B*S, H*D] view (row stride ldv) -> vt [B, H, D, S_pad] in the kernels' key order."""
    S_pad = round_up(S, 64)
    out = torch.empty(B, H, D, S_pad, dtype=v.dtype, device=v.device) if out is None else out
    L.check(L.lib().mh_attn_prep_v(p(v), i64(v.stride(0)), p(out), i32(B), i32(S), i32(H), i32(D), i32(dt_of(v)), _stream()), "mh_attn_prep_v")
    return out


def attn_fwd(q, k, vt, B, S, H, D, causal, seqlens=None, out=None, lse=None):
    """q, k: [B*S, H*D] views (row strides ldq/ldk); vt from attn_prep_v.  Returns (o [B*S, H*D], lse [B,H,S_pad])."""
    out = torch.empty(B * S, H * D, dtype=q.dtype, device=q.device) if out is None else out
    lse = torch.zeros(B, H, round_up(S, 64), dtype=torch.float32, device=q.device) if lse is None else lse
    L.check(L.lib().mh_attn_fwd(p(q), i64(q.stride(0)), p(k), i64(k.stride(0)), p(vt), p(out), i64(out.stride(0)), p(lse),
                                p(seqlens), i32(B), i32(S), i32(H), i32(D), i32(int(causal)), i32(dt_of(q)), _stream()), "mh_attn_fwd")
    return out, lse


def attn_bwd(q, k, v, o, do, lse, B, S, H, D, causal, seqlens=None, dq=None, dk=None, dv=None):
    dq = torch.empty(B * S, H * D, dtype=q.dtype, device=q.device) if dq is None else dq
    dk = torch.empty(B * S, H * D, dtype=q.dtype, device=q.device) if dk is None else dk
    dv = torch.empty(B * S, H * D, dtype=q.dtype, device=q.device) if dv is None else dv
    delta = torch.zeros(2, B, H, round_up(S, 64), dtype=torch.float32, device=q.device)  # [delta | lse*log2e]
    ws = torch.empty(int(L.lib().mh_attn_bwd_ws_elems(i32(B), i32(S), i32(H), i32(D))), dtype=q.dtype, device=q.device)
    L.check(L.lib().mh_attn_bwd(p(q), i64(q.stride(0)), p(k), i64(k.stride(0)), p(v), i64(v.stride(0)), p(o), i64(o.stride(0)),
                                p(do), i64(do.stride(0)), p(lse), p(delta), p(dq), i64(dq.stride(0)), p(dk), i64(dk.stride(0)),
                                p(dv), i64(dv.stride(0)), p(ws), p(seqlens), i32(B), i32(S), i32(H), i32(D), i32(int(causal)),
                                i32(dt_of(q)), _stream()), "mh_attn_bwd")
    return dq, dk, dv


