"""Tensor-level wrappers over the C ABI (include/merlin_hip.h).

PyTorch is used for device memory and streams only: every function checks layouts, allocates
outputs with torch, and enqueues ONE library call on torch's current HIP stream.  No torch
arithmetic happens here, and there is no fallback: without libmerlin_hip.so these raise.
"""
from __future__ import annotations

import os

import torch

from . import _lib as L
from ._lib import C, f32, i32, i64, p, u64

EPI_BIAS, EPI_QUICK_GELU, EPI_RESIDUAL, EPI_ACCUM, EPI_OUT_F32 = L.EPI_BIAS, L.EPI_QUICK_GELU, L.EPI_RESIDUAL, L.EPI_ACCUM, L.EPI_OUT_F32


def dt_of(t_or_dtype) -> int:
    d = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    if d == torch.bfloat16:
        return L.MH_BF16
    if d == torch.float16:
        return L.MH_F16
    if d == torch.float32:
        return L.MH_F32
    raise TypeError(f"unsupported dtype {d}")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- optional per-launch timing (HIP events on the launch stream); used by bench.py's roofline leg ----
_PROF = None


def profile_start():
    global _PROF
    _PROF = []


LAST_PROFILE_BYTES = {}  # {kernel: algorithmic bytes (every operand and the result moved once)} of the last profile_stop()


def profile_stop():
    """-> {kernel: (launches, total_work, total_ms)}; work = FLOPs (gemm/attn) or bytes."""
    global _PROF, LAST_PROFILE_BYTES
    rec, _PROF = _PROF, None
    torch.cuda.synchronize()
    out, nb = {}, {}
    for name, work, s, e, b in rec or []:
        n, w, ms = out.get(name, (0, 0.0, 0.0))
        out[name] = (n + 1, w + work, ms + s.elapsed_time(e))
        nb[name] = nb.get(name, 0.0) + b
    LAST_PROFILE_BYTES = nb
    return out


class _timed:
    __slots__ = ("name", "work", "s", "nbytes")

    def __init__(self, name, work, nbytes=0.0):
        self.name, self.work, self.nbytes = name, work, nbytes

    def __enter__(self):
        if _PROF is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.s.record()

    def __exit__(self, *a):
        if _PROF is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            _PROF.append((self.name, self.work, self.s, e, self.nbytes))


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _rowmajor(t: torch.Tensor):
    assert t.dim() == 2 and t.stride(1) == 1, f"need row-major 2-D, got {tuple(t.shape)} strides {t.stride()}"
    return t.stride(0)


def arch_ok(dev: int = 0) -> bool:
    return bool(L.lib().mh_arch_ok(i32(dev)))


# ---------------------------------------------------------------------------------------------
SKINNY_SPLITK = os.environ.get("MH_SKINNY_SPLITK", "1") != "0"  # A/B switch
SKINNY_SPLITK_ALL = os.environ.get("MH_SKINNY_SPLITK", "1") == "2"  # fp16 too (default: bf16 only, see _skinny_splitk_ok)
SKINNY_MAX_ROWS = 4096


def gemm_nt(a, b, *, out=None, bias=None, resid=None, act=None, accum=False, out_f32=False, n=None, a_t=False, b_t=False):
    """out[M,N] = A @ B^T (+bias)(quick_gelu)(+resid)(+out), 16-bit row-major operands.
    A = a[M,K] (or a[K,M] when a_t: K-strided), B = b[N,K] (or b[K,N] when b_t)."""
    if a_t:
        K, M = a.shape
    else:
        M, K = a.shape
    if b_t:
        assert b.shape[0] == K
        N = b.shape[1] if n is None else n
    else:
        assert b.shape[1] == K
        N = b.shape[0] if n is None else n
    assert a.dtype == b.dtype
    lda, ldb = _rowmajor(a), _rowmajor(b)
    if out is None:
        assert not accum
        out = torch.empty(M, N, dtype=torch.float32 if out_f32 else a.dtype, device=a.device)
    ldc = _rowmajor(out)
    epi = 0
    if bias is not None:
        epi |= EPI_BIAS
    if act == "quick_gelu":
        epi |= EPI_QUICK_GELU
    elif act is not None:
        raise ValueError(act)
    ldr = 0
    if resid is not None:
        epi |= EPI_RESIDUAL
        ldr = _rowmajor(resid)
    if accum:
        epi |= EPI_ACCUM
    if out.dtype == torch.float32:
        epi |= EPI_OUT_F32
    else:
        assert out.dtype == a.dtype
    # short sequences (a single 613-token example, a prompt's prefill): few output tiles and a long contraction -> split K over
    # several blocks per tile, the epilogue applied by the reduce pass (profiles/r03_skinny_gemm.txt)
    splits = 1
    if _skinny_splitk_ok(a.dtype) and M <= SKINNY_MAX_ROWS and N % 4 == 0 and ldc % 4 == 0:
        splits = int(L.lib().mh_gemm_splitk_max(i32(M), i32(N), i32(K)))
    with _timed("gemm_nt", 2.0 * M * N * K, 2.0 * (M * K + N * K) + float(out.element_size()) * M * N):
        if splits > 1:
            ws = _splitk_workspace(a.device, splits * M * N)
            L.check(L.lib().mh_gemm_splitk_epi(p(a), i64(lda), i32(int(a_t)), p(b), i64(ldb), i32(int(b_t)), p(out), i64(ldc), p(bias), p(resid),
                                               i64(ldr), i32(M), i32(N), i32(K), i32(dt_of(a)), i32(epi), i32(splits), p(ws), _stream()),
                    "mh_gemm_splitk_epi")
        else:
            L.check(L.lib().mh_gemm(p(a), i64(lda), i32(int(a_t)), p(b), i64(ldb), i32(int(b_t)), p(out), i64(ldc), p(bias), p(resid),
                                    i64(ldr), i32(M), i32(N), i32(K), i32(dt_of(a)), i32(epi), _stream()), "mh_gemm")
    return out


def gemv(x, w, out=None, resid=None, out_f32=False, n=None):
    """out[M, N] = x[M, K] @ w[N, K]^T (+ resid), <= 16 rows per launch (decode step): weights streamed once per launch."""
    M, K = x.shape
    N = w.shape[0] if n is None else n
    assert w.shape[1] == K and x.dtype == w.dtype
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    step = _gemv_rows_per_launch(K)
    for m0 in range(0, M, step):
        mm = min(step, M - m0)
        xs, os_ = x[m0:m0 + mm], out[m0:m0 + mm]
        rs = resid[m0:m0 + mm] if resid is not None else None
        L.check(L.lib().mh_gemv(p(xs), i64(_rowmajor(xs)), p(w), i64(_rowmajor(w)), p(os_), i64(_rowmajor(os_)), p(rs),
                                i64(_rowmajor(rs) if rs is not None else 0), i32(mm), i32(N), i32(K), i32(dt_of(x)),
                                i32(int(out.dtype == torch.float32)), _stream()), "mh_gemv")
    return out


def gemv_swiglu(x, wgu, out=None):
    """act[M, ff] = silu(x @ Wg^T) * (x @ Wu^T), wgu = [Wg; Wu] [2 ff, K]: the decode step's gate|up projection + SwiGLU in one launch
    (gate / up rounded to 16 bits first, as gemv + swiglu_fwd).  More than 8 rows: the two launches."""
    M, K = x.shape
    ff = wgu.shape[0] // 2
    assert wgu.shape[1] == K and x.dtype == wgu.dtype
    if not _gemv_fused_rows_ok(M, K):
        return swiglu_fwd(gemv(x, wgu), out=out)
    out = torch.empty(M, ff, dtype=x.dtype, device=x.device) if out is None else out
    L.check(L.lib().mh_gemv_swiglu(p(x), i64(_rowmajor(x)), p(wgu), i64(_rowmajor(wgu)), p(out), i64(_rowmajor(out)), i32(M), i32(ff), i32(K),
                                   i32(dt_of(x)), _stream()), "mh_gemv_swiglu")
    return out


FUSED_NORM_MAX_ROWS = 2


def gemv_norm(x, norm_w, eps, w, swiglu=False, out=None):
    """rmsnorm(x; norm_w, eps) @ w^T with the norm done inside the projection's launch (decode step); swiglu=True: w = [Wg; Wu] and the
    result is silu(gate) * up [M, ff].  The fused form pays for 1-2 rows (every block re-normalises the rows: measured slower than the
    separate norm from 4 rows on); otherwise, or with K > 8192: the separate launches."""
    M, K = x.shape
    if M > FUSED_NORM_MAX_ROWS or K > 8192:
        h = rmsnorm_fwd(x, norm_w, eps)
        return gemv_swiglu(h, w, out=out) if swiglu else gemv(h, w, out=out)  # (gemv_swiglu: fused up to 16 rows)
    N = w.shape[0]
    ff = N // 2 if swiglu else 0
    assert w.shape[1] == K and x.dtype == w.dtype == norm_w.dtype and x.is_contiguous()
    out = torch.empty(M, ff if swiglu else N, dtype=x.dtype, device=x.device) if out is None else out
    L.check(L.lib().mh_gemv_norm(p(x), i64(_rowmajor(x)), p(norm_w), f32(eps), p(w), i64(_rowmajor(w)), p(out), i64(_rowmajor(out)), i32(M), i32(N),
                                 i32(K), i32(ff), i32(dt_of(x)), _stream()), "mh_gemv_norm")
    return out


def quant_fp8_b128(w):
    """w [N, K] (16-bit) -> (q uint8 [N, K] OCP e4m3, scales fp32 [N, ceil(K/128)])."""
    N, K = w.shape
    q = torch.empty(N, K, dtype=torch.uint8, device=w.device)
    sc = torch.empty(N, (K + 127) // 128, dtype=torch.float32, device=w.device)
    L.check(L.lib().mh_quant_fp8_b128(p(w), i64(_rowmajor(w)), p(q), p(sc), i32(N), i32(K), i32(dt_of(w)), _stream()), "mh_quant_fp8_b128")
    return q, sc


def gemv_fp8w(x, qw, out=None, resid=None, out_f32=False, n=None):
    """out[M, N] = x[M, K] @ dequant(q, scales)^T (+ resid); qw = (q, scales) from quant_fp8_b128."""
    q, sc = qw
    M, K = x.shape
    N = q.shape[0] if n is None else n
    assert q.shape[1] == K
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    step = _gemv_rows_per_launch(K)
    for m0 in range(0, M, step):
        mm = min(step, M - m0)
        xs, os_ = x[m0:m0 + mm], out[m0:m0 + mm]
        rs = resid[m0:m0 + mm] if resid is not None else None
        L.check(L.lib().mh_gemv_fp8w(p(xs), i64(_rowmajor(xs)), p(q), p(sc), p(os_), i64(_rowmajor(os_)), p(rs),
                                     i64(_rowmajor(rs) if rs is not None else 0), i32(mm), i32(N), i32(K), i32(dt_of(x)),
                                     i32(int(out.dtype == torch.float32)), _stream()), "mh_gemv_fp8w")
    return out


def gemv_fp8w_norm(x, norm_w, eps, qw, swiglu=False, out=None):
    """gemv_norm with fp8 weights (qw = (q, scales) from quant_fp8_b128).  1-2 rows: norm, projection (and SwiGLU) in one launch;
    3-16 rows: separate norm, SwiGLU still fused; more rows: separate launches."""
    q, sc = qw
    M, K = x.shape
    N = q.shape[0]
    ff = N // 2 if swiglu else 0
    if not _gemv_fused_rows_ok(M, K) or K > 8192:
        y = gemv_fp8w(rmsnorm_fwd(x, norm_w, eps), qw)
        return swiglu_fwd(y, out=out) if swiglu else y
    fuse_norm = M <= FUSED_NORM_MAX_ROWS
    if not fuse_norm:
        x = rmsnorm_fwd(x, norm_w, eps)
        if not swiglu:
            return gemv_fp8w(x, qw, out=out)
    assert q.shape[1] == K and x.is_contiguous()
    out = torch.empty(M, ff if swiglu else N, dtype=x.dtype, device=x.device) if out is None else out
    L.check(L.lib().mh_gemv_fp8w_norm(p(x), i64(_rowmajor(x)), p(norm_w if fuse_norm else None), f32(eps), p(q), p(sc), p(out), i64(_rowmajor(out)),
                                      i32(M), i32(N), i32(K), i32(ff), i32(dt_of(x)), _stream()), "mh_gemv_fp8w_norm")
    return out


def gemv_qkv_rope(x, norm_w, eps, w, table, pos, kcache, vcache, H, D, rope_pos=None):
    """Decode-step q|k|v: (input_layernorm +) projection + RoPE at pos + K/V append in one launch; w = 16-bit weight [3HD, K] or the
    (q, scales) pair of quant_fp8_b128.  Returns qkv [M, 3HD] (q, k rotated).  More than 16 rows: the separate launches."""
    M, K = x.shape
    fp8 = isinstance(w, (tuple, list))
    if not _gemv_fused_rows_ok(M, K) or K > 8192 or (M > 8 and (D // 2) % 16):
        h = rmsnorm_fwd(x, norm_w, eps)
        qkv = gemv_fp8w(h, w) if fp8 else gemv(h, w)
        decode_rope_append(qkv, table, pos, kcache, vcache, H, D, rope_pos=rope_pos)
        return qkv
    if M > FUSED_NORM_MAX_ROWS:
        x, norm_w = rmsnorm_fwd(x, norm_w, eps), None
    assert x.is_contiguous() and pos.dtype == torch.int32 and kcache.is_contiguous() and vcache.is_contiguous()
    qkv = torch.empty(M, 3 * H * D, dtype=x.dtype, device=x.device)
    wq, sc = (w if fp8 else (None, None))
    L.check(L.lib().mh_gemv_qkv_rope(p(x), i64(_rowmajor(x)), p(norm_w), f32(eps), p(None if fp8 else w), i64(0 if fp8 else _rowmajor(w)), p(wq), p(sc),
                                     p(qkv), i64(_rowmajor(qkv)), i32(M), i32(K), i32(dt_of(x)), p(table), p(pos), p(rope_pos), p(kcache), p(vcache),
                                     i32(H), i32(D), i32(kcache.shape[1]), _stream()), "mh_gemv_qkv_rope")
    return qkv


def decode_rope_append(qkv, table, pos, kcache, vcache, H, D, rope_pos=None):
    """qkv [B, 3*H*D] of the new tokens (rotated in place at rope_pos[b], default pos[b]); k, v appended to kcache/vcache [B, Smax, H*D]
    at row pos[b]."""
    B = qkv.shape[0]
    assert qkv.is_contiguous() and pos.dtype == torch.int32 and kcache.is_contiguous() and vcache.is_contiguous()
    assert rope_pos is None or rope_pos.dtype == torch.int32
    L.check(L.lib().mh_decode_rope_append(p(qkv), p(table), p(pos), p(rope_pos), p(kcache), p(vcache), i32(B), i32(H), i32(D),
                                          i32(kcache.shape[1]), i32(dt_of(qkv)), _stream()), "mh_decode_rope_append")


def attn_decode(q, kcache, vcache, lens, H, D, out=None, split_kv=True):
    """q [B, H*D] view (row stride ldq) against the cache [B, Smax, H*D]; keys [0, lens[b])."""
    B = q.shape[0]
    out = torch.empty(B, H * D, dtype=q.dtype, device=q.device) if out is None else out
    Smax = kcache.shape[1]
    splits = int(L.lib().mh_attn_decode_splits(i32(B), i32(H), i32(Smax)))
    ws = torch.empty(B * H * splits * (D + 2), dtype=torch.float32, device=q.device) if (splits > 1 and split_kv) else None
    L.check(L.lib().mh_attn_decode(p(q), i64(q.stride(0)), p(kcache), p(vcache), p(out), p(lens), i32(B), i32(H), i32(D),
                                   i32(Smax), p(ws), i32(dt_of(q)), _stream()), "mh_attn_decode")
    return out


def attn_decode_fused_merge(on: bool):
    """A/B switch: split-KV partials merged by a second launch (default) or by the last block of a (b, h) to finish."""
    L.lib().mh_attn_decode_fused_merge(i32(1 if on else 0))


_splitk_ws = {}
SPLITK_WS_MIN = 256 * 65536  # floats: one round of 256 tiles of fp32 partials (the most mh_gemm_splitk_max ever asks for)


def _splitk_workspace(device, numel):
    """fp32 partials of the split-K GEMMs: ONE grow-only buffer per (device, stream).  Launches on one stream are ordered, so
    consecutive GEMMs may share it; two streams never do (a size-keyed, process-global cache let concurrent streams race on the
    partials and thrashed on variable-length batches).  While the stream is being captured into a HIP graph the buffer comes from
    the graph's own memory pool and is NOT cached (it must not outlive the graph; the pool re-uses the block for the next GEMM of
    the capture): a captured prefill / wide decode step keeps the split path and with it the eager path's summation order -
    graph and eager logits stay bit-identical."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(numel, dtype=torch.float32, device=device)
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _splitk_ws.get(key)
    if ws is None or ws.numel() < numel:
        if len(_splitk_ws) > 16:  # streams come and go (side streams of captures, tests): drop the buffers of the others
            _splitk_ws.clear()
        ws = _splitk_ws[key] = torch.empty(max(numel, SPLITK_WS_MIN), dtype=torch.float32, device=device)
    return ws


def _skinny_splitk_ok(dtype):
    """Split-K for skinny products (profiles/r03_skinny_gemm.txt).  Kept to bf16, the performance dtype: fp16 is the dtype BASELINE's
    logits tolerance is stated in, and its full-depth bound (tests/test_model_gpu.py) is held with the one-pass summation order it was
    measured with - the single-element maximum moves by +-25 % under ANY change of summation order.  MH_SKINNY_SPLITK=0 / =2 force it
    off / on for every dtype (A/B).  The same decision inside and outside a HIP-graph capture (_splitk_workspace)."""
    if not SKINNY_SPLITK:
        return False
    return dtype == torch.bfloat16 or SKINNY_SPLITK_ALL


_tail_plans = {}


def _tail_plan(M, N, T):
    """256 x 256 output tiles on 256 CUs run in rounds; a weight gradient whose tile count is not a multiple of 256 leaves the
    last round partly empty (gate|up: 86 x 16 = 1376 tiles = 5.375 rounds -> 6; down: 16 x 43 = 688 = 2.69 -> 3: 10-12 % of
    GEMMs that contract over all 32 768 tokens).  Plan: cut the output into a part that fills whole rounds and a remainder that
    is split along K (tokens) so that ITS blocks fill whole, proportionally shorter rounds.  Returns None or
    (axis 'm' | 'n', cut in elements, splits of the remainder)."""
    key = (M, N, T)
    if key in _tail_plans:
        return _tail_plans[key]
    if os.environ.get("MH_NO_TAIL_PLAN"):  # A/B switch for kernel development
        return None
    tm, tn = (M + 255) // 256, (N + 255) // 256
    tiles, nk = tm * tn, (T + 63) // 64
    plan = None
    if tiles >= 256 and nk >= 128 and M % 256 == 0 and N % 256 == 0:
        best = -(-tiles // 256) * 0.96  # must beat the plain launch by > 4 % (the reduce pass and a second launch are not free)
        for axis, ta, tb in (("m", tm, tn), ("n", tn, tm)):
            for a in range(ta - 1, 0, -1):
                ca = -(-(a * tb) // 256)
                if ca * 256 - a * tb > 0.03 * a * tb:
                    continue
                rem = (ta - a) * tb
                for sp in range(2, 17):
                    if nk // sp < 16:
                        break
                    # remainder rounds are 1/sp long; its fp32 partials (sp x rem tiles) are written and read once at ~5 TB/s,
                    # expressed in rounds (one full-K round of 256 tiles takes ~T x 25.7 ns at the kernel's rate)
                    cost = ca + -(-(rem * sp) // 256) / sp + 0.02 + sp * rem * 524288 / 5e12 / (T * 25.7e-9)
                    if cost < best:
                        best, plan = cost, (axis, a * 256, sp)
    _tail_plans[key] = plan
    return plan


def _wgrad_call(dy, x, out, accum, splits):
    T, M = dy.shape
    N = x.shape[1]
    ws = _splitk_workspace(dy.device, splits * M * N) if splits > 1 else None
    L.check(L.lib().mh_gemm_splitk(p(dy), i64(_rowmajor(dy)), i32(1), p(x), i64(_rowmajor(x)), i32(1), p(out), i64(_rowmajor(out)), i32(M), i32(N),
                                   i32(T), i32(dt_of(dy)), i32(int(accum)), i32(int(out.dtype == torch.float32)), i32(splits), p(ws),
                                   _stream()), "mh_gemm_splitk")


def wgrad_tn(dy, x, out, accum):
    """out[N_out, K_in] (+)= dy[T, N_out]^T @ x[T, K_in]: both operands K-strided as they lie in memory, any T (the
    kernel reads rows >= T of the last K-tile as zeros), split-K when the output has too few tiles to fill the chip, and a
    two-part launch (full rounds + a K-split remainder, see _tail_plan) when its tile count leaves a partly empty last round."""
    T, M = dy.shape
    T2, N = x.shape
    assert T == T2 and dy.dtype == x.dtype and M % 8 == 0 and N % 8 == 0
    with _timed("gemm_nt", 2.0 * M * N * T, 2.0 * (M * T + N * T + M * N)):
        plan = _tail_plan(M, N, T)
        if plan is None:
            _wgrad_call(dy, x, out, accum, int(L.lib().mh_gemm_splitk_max(i32(M), i32(N), i32(T))))
        else:
            axis, cut, sp = plan
            if axis == "m":
                _wgrad_call(dy[:, :cut], x, out[:cut], accum, 1)
                _wgrad_call(dy[:, cut:], x, out[cut:], accum, sp)
            else:
                _wgrad_call(dy, x[:, :cut], out[:, :cut], accum, 1)
                _wgrad_call(dy, x[:, cut:], out[:, cut:], accum, sp)
    return out


class _WgradProblem(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("lddy", C.c_int64), ("x", C.c_void_p), ("ldx", C.c_int64), ("out", C.c_void_p), ("ldo", C.c_int64),
                ("M", C.c_int), ("N", C.c_int), ("accumulate", C.c_int), ("reserved", C.c_int)]


WGRAD_GROUPED = os.environ.get("MH_WGRAD_GROUPED", "1") != "0"  # A/B switch
WGRAD_GROUPED_MIN_TILES = 32  # below that (the tiny test models) the per-problem launches are used unless forced


def wgrad_group_pays(T, shapes):
    """Launch-plan predicate of wgrad_tn_grouped for problems of shapes [(M_p, N_p)] over T tokens (host arithmetic only)."""
    tiles = sum(((m + 255) // 256) * ((n + 255) // 256) for m, n in shapes)
    return WGRAD_GROUPED and len(shapes) <= 8 and T >= 1024 and tiles >= WGRAD_GROUPED_MIN_TILES and all(m % 8 == 0 and n % 8 == 0 for m, n in shapes)


def wgrad_tn_grouped(problems, accum, force=False):
    """[(dy [T, M_p], x [T, N_p], out [M_p, N_p])] -> out_p (+)= dy_p^T x_p for all p in ONE launch (mh_wgrad_grouped: one block per
    256 x 256 tile of any problem, whole contraction, no fp32 partials); falls back to wgrad_tn per problem when the group is too small
    to matter or a problem does not meet the kernel's layout conditions."""
    T = problems[0][0].shape[0]
    ok = len(problems) <= 8 and all(dy.shape[0] == T and x.shape[0] == T and dy.dtype == x.dtype == out.dtype and
                                    out.dtype in (torch.bfloat16, torch.float16) for dy, x, out in problems)
    if ok and (force or wgrad_group_pays(T, [(dy.shape[1], x.shape[1]) for dy, x, _ in problems])):
        arr = (_WgradProblem * len(problems))()
        flops = nbytes = 0.0
        for i, (dy, x, out) in enumerate(problems):
            arr[i] = _WgradProblem(dy.data_ptr(), _rowmajor(dy), x.data_ptr(), _rowmajor(x), out.data_ptr(), _rowmajor(out), dy.shape[1], x.shape[1],
                                   int(accum), 0)
            flops += 2.0 * dy.shape[1] * x.shape[1] * T
            nbytes += 2.0 * (dy.shape[1] * T + x.shape[1] * T + dy.shape[1] * x.shape[1])
        with _timed("gemm_nt", flops, nbytes):
            rc = L.lib().mh_wgrad_grouped(arr, i32(len(problems)), i32(T), i32(dt_of(problems[0][0])), _stream())
        if rc == 0:
            return
        if rc != -4:  # MH_ERR_SHAPE = "not for this kernel": per-problem launches below; anything else is an error
            L.check(rc, "mh_wgrad_grouped")
    for dy, x, out in problems:
        wgrad_tn(dy, x, out, accum)


def gemm_nt_rope(a, b, table, S, H, D, out=None):
    """qkv = a @ b^T with RoPE applied to the q and k heads in the GEMM epilogue (b = fused [q; k; v] weight, 3*H*D rows)."""
    M, K = a.shape
    N = b.shape[0]
    assert N == 3 * H * D and b.shape[1] == K and a.dtype == b.dtype
    out = torch.empty(M, N, dtype=a.dtype, device=a.device) if out is None else out
    with _timed("gemm_nt", 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N)):
        L.check(L.lib().mh_gemm_nt_rope(p(a), i64(_rowmajor(a)), p(b), i64(_rowmajor(b)), p(out), i64(_rowmajor(out)), i32(M), i32(N),
                                        i32(K), i32(dt_of(a)), p(table), i32(S), i32(D), i32(2 * H * D), _stream()), "mh_gemm_nt_rope")
    return out


def gemm_swiglu_fwd(x, wgu):
    """gu = x @ wgu^T ([gate; up] rows) and act = silu(gate) * up from ONE GEMM launch.  Returns (gu, act)."""
    M, K = x.shape
    ff = wgu.shape[0] // 2
    gu = torch.empty(M, 2 * ff, dtype=x.dtype, device=x.device)
    act = torch.empty(M, ff, dtype=x.dtype, device=x.device)
    with _timed("gemm_nt", 2.0 * M * 2 * ff * K, 2.0 * (M * K + 2 * ff * K + 3 * M * ff)):
        L.check(L.lib().mh_gemm_swiglu_fwd(p(x), i64(_rowmajor(x)), p(wgu), i64(_rowmajor(wgu)), p(gu), i64(2 * ff), p(act), i64(ff),
                                           i32(M), i32(ff), i32(K), i32(dt_of(x)), _stream()), "mh_gemm_swiglu_fwd")
    return gu, act


def gemm_swiglu_bwd(dy, wd, gu):
    """dgu = swiglu_bwd(gu, dy @ wd) with the SwiGLU backward in the dgrad GEMM's epilogue (dact never stored).
    wd = down_proj.weight [d_model, ff]."""
    M, K = dy.shape
    ff = wd.shape[1]
    assert wd.shape[0] == K and gu.shape == (M, 2 * ff)
    dgu = torch.empty_like(gu)
    with _timed("gemm_nt", 2.0 * M * ff * K, 2.0 * (M * K + ff * K + 4 * M * ff)):
        L.check(L.lib().mh_gemm_swiglu_bwd(p(dy), i64(_rowmajor(dy)), p(wd), i64(_rowmajor(wd)), p(gu), i64(_rowmajor(gu)), p(dgu),
                                           i64(_rowmajor(dgu)), i32(M), i32(ff), i32(K), i32(dt_of(dy)), _stream()), "mh_gemm_swiglu_bwd")
    return dgu


def gemm_gelu_fwd(x, w1, bias):
    """(f1, a) = (x @ w1^T + bias, quick_gelu(f1)) from ONE GEMM launch (CLIP MLP fc1; the backward needs f1, fc2 needs a)."""
    M, K = x.shape
    N = w1.shape[0]
    f1 = torch.empty(M, N, dtype=x.dtype, device=x.device)
    a = torch.empty(M, N, dtype=x.dtype, device=x.device)
    with _timed("gemm_nt", 2.0 * M * N * K, 2.0 * (M * K + N * K + 2 * M * N)):
        L.check(L.lib().mh_gemm_gelu_fwd(p(x), i64(_rowmajor(x)), p(w1), i64(_rowmajor(w1)), p(bias), p(f1), i64(N), p(a), i64(N), i32(M), i32(N), i32(K),
                                         i32(dt_of(x)), _stream()), "mh_gemm_gelu_fwd")
    return f1, a


def gemm_gelu_bwd(dy, w2, f1):
    """df1 = quick_gelu'(f1) * (dy @ w2) with the GELU backward in the dgrad GEMM's store phase (dy @ w2 never stored); w2 = fc2.weight [d, ff]."""
    M, K = dy.shape
    N = w2.shape[1]
    assert w2.shape[0] == K and f1.shape == (M, N)
    df1 = torch.empty_like(f1)
    with _timed("gemm_nt", 2.0 * M * N * K, 2.0 * (M * K + N * K + 2 * M * N)):
        L.check(L.lib().mh_gemm_gelu_bwd(p(dy), i64(_rowmajor(dy)), p(w2), i64(_rowmajor(w2)), p(f1), i64(_rowmajor(f1)), p(df1), i64(_rowmajor(df1)),
                                         i32(M), i32(N), i32(K), i32(dt_of(dy)), _stream()), "mh_gemm_gelu_bwd")
    return df1


def quant_fp8_rows(x, k_pad=None):
    """x [R, K] (16-bit) -> (q uint8 [R, K] OCP e4m3, scales fp32 [R]): one scale per row.  k_pad > K: q is [R, k_pad], zero behind K."""
    R, K = x.shape
    q = torch.zeros(R, k_pad, dtype=torch.uint8, device=x.device) if (k_pad is not None and k_pad > K) else torch.empty(R, K, dtype=torch.uint8, device=x.device)
    sc = torch.empty(R, dtype=torch.float32, device=x.device)
    L.check(L.lib().mh_quant_fp8_rows(p(x), i64(_rowmajor(x)), p(q), i64(q.stride(0)), p(sc), i32(R), i32(K), i32(dt_of(x)), _stream()), "mh_quant_fp8_rows")
    return q, sc


_amax_ws = {}


def _b3(b8):
    """(q, row scales[, block exponents]) -> always a triple."""
    return (b8[0], b8[1], b8[2] if len(b8) > 2 else None)


def _exp_image(n_rows, k, device):
    G = round_up((k // 128) * 64, 4096)
    return torch.zeros(16 + ((n_rows + 255) // 256) * 2 * G, dtype=torch.uint8, device=device)  # 16-byte flag header + image


def quant_fp8_rows_e4(w):
    """Weights for the fp8 GEMMs: w [N, K] -> (q uint8 [N, K] e4m3, s fp32 [N], exps): per-row scale s[n] and a 4-bit exponent per
    (row, 128-k block), block scale = s[n] * 2^-e - BASELINE cfg 5's per-128-block scales in the form the MFMA applies itself."""
    N, K = w.shape
    assert K % 128 == 0 and w.stride(1) == 1
    q = torch.empty(N, K, dtype=torch.uint8, device=w.device)
    sc = torch.empty(N, dtype=torch.float32, device=w.device)
    ex = _exp_image(N, K, w.device)
    L.check(L.lib().mh_quant_fp8_rows_e4(p(w), i64(_rowmajor(w)), p(q), p(sc), p(ex), i32(N), i32(K), i32(dt_of(w)), _stream()), "mh_quant_fp8_rows_e4")
    return q, sc, ex


def quant_fp8_rows_t_e4(w):
    """w [R, C] -> (qt [C, round_up(R, 128)], s [C], exps): the same format for w^T (rows = input channels, blocks of 128 output channels)."""
    R, C_ = w.shape
    Rp = round_up(R, 128)
    qt = torch.empty(C_, Rp, dtype=torch.uint8, device=w.device)
    sc = torch.empty(C_, dtype=torch.float32, device=w.device)
    ex = _exp_image(C_, Rp, w.device)
    ws = _amax_ws.get((w.device, C_))
    if ws is None:
        ws = _amax_ws[(w.device, C_)] = torch.empty(C_, dtype=torch.int32, device=w.device)
    L.check(L.lib().mh_quant_fp8_rows_t_e4(p(w), i64(_rowmajor(w)), p(qt), i64(Rp), p(sc), p(ex), p(ws), i32(R), i32(C_), i32(dt_of(w)), _stream()),
            "mh_quant_fp8_rows_t_e4")
    return qt, sc, ex


def quant_fp8_rows_t(x):
    """x [R, C] (16-bit) -> (qt uint8 [C, round_up(R, 128)] = e4m3(x^T / s), s fp32 [C]): the column-scaled TRANSPOSED operand
    (zero-filled pad columns) the wgrad / dgrad GEMMs of the fp8 training step contract over."""
    R, C_ = x.shape
    Rp = round_up(R, 128)
    qt = torch.empty(C_, Rp, dtype=torch.uint8, device=x.device)
    sc = torch.empty(C_, dtype=torch.float32, device=x.device)
    ws = _amax_ws.get((x.device, C_))
    if ws is None:
        ws = _amax_ws[(x.device, C_)] = torch.empty(C_, dtype=torch.int32, device=x.device)
    L.check(L.lib().mh_quant_fp8_rows_t(p(x), i64(_rowmajor(x)), p(qt), i64(Rp), p(sc), p(ws), i32(R), i32(C_), i32(dt_of(x)), _stream()),
            "mh_quant_fp8_rows_t")
    return qt, sc


_both_ws = {}


def quant_fp8_both(x, c_pad=None, amax=None):
    """x [R, C] (16-bit) -> ((q [R, C], s_row [R]), (qt [C, round_up(R, 128)], s_col [C])): the row-quantised operand (dgrad) and the
    transposed, per-feature-scaled operand (wgrad) of a gradient tensor from two reads of it.  c_pad > C: q is [R, c_pad] with zero
    columns behind C (a contraction length that is not a whole number of 128-blocks: the vocabulary).
    amax: int32 [R + C] = the tensor's row then column maxima as its producer left them (gemm_fp8_swiglu_bwd(..., want_amax=True)):
    ONE read of x, identical bytes."""
    R, C_ = x.shape
    Rp = round_up(R, 128)
    if c_pad is not None and c_pad > C_:
        q = torch.zeros(R, c_pad, dtype=torch.uint8, device=x.device)
    else:
        q = torch.empty(R, C_, dtype=torch.uint8, device=x.device)
    sr = torch.empty(R, dtype=torch.float32, device=x.device)
    qt = torch.empty(C_, Rp, dtype=torch.uint8, device=x.device)
    sc = torch.empty(C_, dtype=torch.float32, device=x.device)
    if amax is not None:
        assert amax.dtype == torch.int32 and amax.numel() == R + C_ and amax.is_contiguous()
        L.check(L.lib().mh_quant_fp8_rows_and_t_pre(p(x), i64(_rowmajor(x)), p(q), i64(q.stride(0)), p(sr), p(qt), i64(Rp), p(sc), p(amax), i32(R), i32(C_),
                                                    i32(dt_of(x)), _stream()), "mh_quant_fp8_rows_and_t_pre")
        return (q, sr), (qt, sc)
    ws = _both_ws.get((x.device, R + C_))
    if ws is None:
        if len(_both_ws) > 16:
            _both_ws.clear()
        ws = _both_ws[(x.device, R + C_)] = torch.empty(R + C_, dtype=torch.int32, device=x.device)
    L.check(L.lib().mh_quant_fp8_rows_and_t(p(x), i64(_rowmajor(x)), p(q), i64(q.stride(0)), p(sr), p(qt), i64(Rp), p(sc), p(ws), i32(R), i32(C_),
                                            i32(dt_of(x)), _stream()), "mh_quant_fp8_rows_and_t")
    return (q, sr), (qt, sc)


def quant_fp8_t_from_rows(x, row_scales):
    """x [R, C] (16-bit) -> (qt uint8 [C, round_up(R, 128)], s fp32 [C] = one tensor-wide scale, the largest of x's row scales
    as produced by quant_fp8_rows): the transposed operand of the wgrad GEMMs in ONE pass over x."""
    R, C_ = x.shape
    Rp = round_up(R, 128)
    qt = torch.empty(C_, Rp, dtype=torch.uint8, device=x.device)
    sc = torch.empty(C_, dtype=torch.float32, device=x.device)
    L.check(L.lib().mh_max_to_vec(p(row_scales), i32(row_scales.numel()), p(sc), i32(C_), _stream()), "mh_max_to_vec")
    L.check(L.lib().mh_quant_fp8_t_scaled(p(x), i64(_rowmajor(x)), p(qt), i64(Rp), p(sc), i32(R), i32(C_), i32(dt_of(x)), _stream()), "mh_quant_fp8_t_scaled")
    return qt, sc


def gemm_fp8_swiglu_bwd(dy8, wdt8, gu, want_amax=False):
    """dgu = swiglu_bwd(gu, dy Wd) on the scaled-fp8 MFMA; dy8 = rowquant(dy) [T, d], wdt8 = rowquant(Wd^T) [ff, d].
    want_amax: returns (dgu, amax int32 [T + 2 ff]) - the row / column maxima of |dgu| taken in the GEMM's store phase (quant_fp8_both(dgu, amax=...))."""
    (qa, sa), (qb, sb, eb) = dy8[:2], _b3(wdt8)
    M, K = qa.shape
    ff = qb.shape[0]
    assert qb.shape[1] == K and gu.shape == (M, 2 * ff)
    dgu = torch.empty_like(gu)
    if want_amax:
        amax = torch.empty(M + 2 * ff, dtype=torch.int32, device=gu.device)
        with _timed("gemm_fp8", 2.0 * M * ff * K):
            L.check(L.lib().mh_gemm_fp8_swiglu_bwd_amax(p(qa), i64(qa.stride(0)), p(sa), p(qb), i64(qb.stride(0)), p(sb), p(eb), p(gu), i64(_rowmajor(gu)),
                                                        p(dgu), i64(_rowmajor(dgu)), p(amax), i32(M), i32(ff), i32(K), i32(dt_of(gu)), _stream()),
                    "mh_gemm_fp8_swiglu_bwd_amax")
        return dgu, amax
    with _timed("gemm_fp8", 2.0 * M * ff * K):
        L.check(L.lib().mh_gemm_fp8_swiglu_bwd(p(qa), i64(qa.stride(0)), p(sa), p(qb), i64(qb.stride(0)), p(sb), p(eb), p(gu), i64(_rowmajor(gu)),
                                               p(dgu), i64(_rowmajor(dgu)), i32(M), i32(ff), i32(K), i32(dt_of(gu)), _stream()), "mh_gemm_fp8_swiglu_bwd")
    return dgu


def gemm_fp8(a8, b8, out_dtype=torch.bfloat16, out=None, bias=None, resid=None, act=None, accum=False, dt16=torch.bfloat16):
    """out[M, N] = (sa qa) @ (sb qb)^T on the scaled-fp8 MFMA; a8 = (qa [M, K] uint8, sa [M]), b8 = (qb [N, K], sb [N]).
    A float32 `out` / out_dtype selects the fp32 store phase (the lm_head logits); dt16 is then the type of bias / resid."""
    (qa, sa), (qb, sb, eb) = a8[:2], _b3(b8)
    M, K = qa.shape
    N = qb.shape[0]
    assert qb.shape[1] == K
    out = torch.empty(M, N, dtype=out_dtype, device=qa.device) if out is None else out
    epi = 0
    if bias is not None:
        epi |= EPI_BIAS
    if act == "quick_gelu":
        epi |= EPI_QUICK_GELU
    ldr = 0
    if resid is not None:
        epi |= EPI_RESIDUAL
        ldr = _rowmajor(resid)
    if accum:
        assert out is not None
        epi |= EPI_ACCUM
    dt = dt_of(out)
    if out.dtype == torch.float32:
        epi |= EPI_OUT_F32
        dt = dt_of(dt16)
    with _timed("gemm_fp8", 2.0 * M * N * K):
        L.check(L.lib().mh_gemm_fp8(p(qa), i64(qa.stride(0)), p(sa), p(qb), i64(qb.stride(0)), p(sb), p(eb), p(out), i64(_rowmajor(out)), p(bias),
                                    p(resid), i64(ldr), i32(M), i32(N), i32(K), i32(dt), i32(epi), _stream()), "mh_gemm_fp8")
    return out


def gemm_fp8_rope(a8, b8, table, S, H, D, out_dtype=torch.bfloat16):
    """fp8 q|k|v projection with RoPE in the epilogue (see gemm_nt_rope)."""
    (qa, sa), (qb, sb, eb) = a8[:2], _b3(b8)
    M, K = qa.shape
    N = qb.shape[0]
    out = torch.empty(M, N, dtype=out_dtype, device=qa.device)
    with _timed("gemm_fp8", 2.0 * M * N * K):
        L.check(L.lib().mh_gemm_fp8_rope(p(qa), i64(qa.stride(0)), p(sa), p(qb), i64(qb.stride(0)), p(sb), p(eb), p(out), i64(N), i32(M), i32(N),
                                         i32(K), i32(dt_of(out)), p(table), i32(S), i32(D), i32(2 * H * D), _stream()), "mh_gemm_fp8_rope")
    return out


def gemm_fp8_swiglu_fwd(a8, b8, out_dtype=torch.bfloat16):
    """fp8 gate|up projection with SwiGLU in the epilogue (see gemm_swiglu_fwd).  Returns (gu, act)."""
    (qa, sa), (qb, sb, eb) = a8[:2], _b3(b8)
    M, K = qa.shape
    ff = qb.shape[0] // 2
    gu = torch.empty(M, 2 * ff, dtype=out_dtype, device=qa.device)
    act = torch.empty(M, ff, dtype=out_dtype, device=qa.device)
    with _timed("gemm_fp8", 2.0 * M * 2 * ff * K):
        L.check(L.lib().mh_gemm_fp8_swiglu_fwd(p(qa), i64(qa.stride(0)), p(sa), p(qb), i64(qb.stride(0)), p(sb), p(eb), p(gu), i64(2 * ff), p(act),
                                               i64(ff), i32(M), i32(ff), i32(K), i32(dt_of(gu)), _stream()), "mh_gemm_fp8_swiglu_fwd")
    return gu, act


def transpose16(x, r_pad=None, out=None):
    """x[R, C] (16-bit) -> out[C, R_pad] with zero-filled tail columns."""
    R, Cc = x.shape
    ldi = _rowmajor(x)
    r_pad = R if r_pad is None else r_pad
    if out is None:
        out = torch.empty(Cc, r_pad, dtype=x.dtype, device=x.device)
    L.check(L.lib().mh_transpose16(p(x), i64(ldi), p(out), i64(_rowmajor(out)), i32(R), i32(Cc), i32(r_pad), _stream()), "mh_transpose16")
    return out


def rmsnorm_fwd(x, w, eps, out=None):
    rows, d = x.shape
    assert x.is_contiguous() and w.is_contiguous()
    out = torch.empty_like(x) if out is None else out
    L.check(L.lib().mh_rmsnorm_fwd(p(x), p(w), p(out), p(None), i32(rows), i32(d), f32(eps), i32(dt_of(x)), _stream()), "mh_rmsnorm_fwd")
    return out


def rmsnorm_fwd_q8(x, w, eps):
    """-> (y, (q uint8 [rows, d], scales fp32 [rows])): RMSNorm + row-quantised e4m3 copy of its output in one launch."""
    rows, d = x.shape
    assert x.is_contiguous() and w.is_contiguous()
    y = torch.empty_like(x)
    q = torch.empty(rows, d, dtype=torch.uint8, device=x.device)
    sc = torch.empty(rows, dtype=torch.float32, device=x.device)
    L.check(L.lib().mh_rmsnorm_fwd_q8(p(x), p(w), p(y), p(q), p(sc), i32(rows), i32(d), f32(eps), i32(dt_of(x)), _stream()), "mh_rmsnorm_fwd_q8")
    return y, (q, sc)


def norm_partials(rows: int) -> int:
    return int(L.lib().mh_norm_bwd_partials(i32(rows)))


def reduce_partials(partial, nblk, d, out, accumulate):
    L.check(L.lib().mh_reduce_partials(p(partial), i32(nblk), i32(d), p(out), i32(dt_of(out)), i32(int(accumulate)), _stream()), "mh_reduce_partials")


def rmsnorm_bwd(x, w, dy, eps, dx=None, accumulate_dx=False, dw_out=None, dw_accumulate=False, ws=None):
    """Returns dx; writes (or accumulates) dw into dw_out (16-bit or fp32 [d])."""
    rows, d = x.shape
    nblk = norm_partials(rows)
    part = torch.empty(nblk, d, dtype=torch.float32, device=x.device) if ws is None else ws
    dx = torch.empty_like(x) if dx is None else dx
    L.check(L.lib().mh_rmsnorm_bwd(p(x), p(w), p(dy), p(dx), p(part), i32(rows), i32(d), f32(eps), i32(dt_of(x)),
                                   i32(int(accumulate_dx)), _stream()), "mh_rmsnorm_bwd")
    if dw_out is not None:
        reduce_partials(part, nblk, d, dw_out, dw_accumulate)
    return dx


def layernorm_fwd(x, w, b, eps, out=None):
    rows, d = x.shape
    assert x.is_contiguous()
    out = torch.empty_like(x) if out is None else out
    L.check(L.lib().mh_layernorm_fwd(p(x), p(w), p(b), p(out), i32(rows), i32(d), f32(eps), i32(dt_of(x)), _stream()), "mh_layernorm_fwd")
    return out


def layernorm_bwd(x, w, dy, eps, dx=None, accumulate_dx=False, dw_out=None, db_out=None, accumulate=False):
    rows, d = x.shape
    nblk = norm_partials(rows)
    pw = torch.empty(nblk, d, dtype=torch.float32, device=x.device)
    pb = torch.empty(nblk, d, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x) if dx is None else dx
    L.check(L.lib().mh_layernorm_bwd(p(x), p(w), p(dy), p(dx), p(pw), p(pb), i32(rows), i32(d), f32(eps), i32(dt_of(x)),
                                     i32(int(accumulate_dx)), _stream()), "mh_layernorm_bwd")
    if dw_out is not None:
        reduce_partials(pw, nblk, d, dw_out, accumulate)
    if db_out is not None:
        reduce_partials(pb, nblk, d, db_out, accumulate)
    return dx


def colsum(x, out, accumulate=False):
    """out[d] (+)= sum over rows of x[rows, d] (bias gradient)."""
    rows, d = x.shape
    nblk = norm_partials(rows)
    part = torch.empty(nblk, d, dtype=torch.float32, device=x.device)
    L.check(L.lib().mh_colsum_partial(p(x), i64(_rowmajor(x)), p(part), i32(rows), i32(d), i32(dt_of(x)), _stream()), "mh_colsum_partial")
    reduce_partials(part, nblk, d, out, accumulate)
    return out


def swiglu_fwd(gu, out=None):
    rows, ff2 = gu.shape
    ff = ff2 // 2
    assert gu.is_contiguous()
    out = torch.empty(rows, ff, dtype=gu.dtype, device=gu.device) if out is None else out
    L.check(L.lib().mh_swiglu_fwd(p(gu), p(out), i32(rows), i32(ff), i32(dt_of(gu)), _stream()), "mh_swiglu_fwd")
    return out


def swiglu_bwd(gu, dout, dgu=None):
    rows, ff2 = gu.shape
    dgu = torch.empty_like(gu) if dgu is None else dgu
    L.check(L.lib().mh_swiglu_bwd(p(gu), p(dout), p(dgu), i32(rows), i32(ff2 // 2), i32(dt_of(gu)), _stream()), "mh_swiglu_bwd")
    return dgu


def quick_gelu_fwd(x, out=None):
    out = torch.empty_like(x) if out is None else out
    L.check(L.lib().mh_quick_gelu_fwd(p(x), p(out), i64(x.numel()), i32(dt_of(x)), _stream()), "mh_quick_gelu_fwd")
    return out


def quick_gelu_bwd(x, dy, dx=None):
    dx = torch.empty_like(x) if dx is None else dx
    L.check(L.lib().mh_quick_gelu_bwd(p(x), p(dy), p(dx), i64(x.numel()), i32(dt_of(x)), _stream()), "mh_quick_gelu_bwd")
    return dx


def add(a, b, out=None):
    out = torch.empty_like(a) if out is None else out
    L.check(L.lib().mh_add(p(a), p(b), p(out), i64(a.numel()), i32(dt_of(a)), _stream()), "mh_add")
    return out


def convert(src, dst):
    assert src.numel() == dst.numel() and src.is_contiguous() and dst.is_contiguous()
    L.check(L.lib().mh_convert(p(src), i32(dt_of(src)), p(dst), i32(dt_of(dst)), i64(src.numel()), _stream()), "mh_convert")
    return dst


def fill_normal_(t, key: int, start: int = 0, sigma: float = 0.02, offset: float = 0.0):
    assert t.is_contiguous()
    L.check(L.lib().mh_fill_normal(p(t), i64(t.numel()), u64(key), i64(start), f32(sigma), f32(offset), i32(dt_of(t)), _stream()), "mh_fill_normal")
    return t


def rope_table(S, D, theta, device):
    tab = torch.empty(S, D // 2, 2, dtype=torch.float32, device=device)
    L.check(L.lib().mh_rope_table(p(tab), i32(S), i32(D), f32(theta), _stream()), "mh_rope_table")
    return tab


def rope_qk_(qkv, table, S, H, D, inverse=False):
    """qkv [T, 3*H*D] (fused q|k|v rows) rotated in place on q and k."""
    T = qkv.shape[0]
    assert qkv.is_contiguous() and qkv.shape[1] == 3 * H * D
    L.check(L.lib().mh_rope_qk(p(qkv), p(table), i32(T), i32(S), i32(H), i32(D), i32(int(inverse)), i32(dt_of(qkv)), _stream()), "mh_rope_qk")
    return qkv


def attn_fwd2(q, k, v, B, S, H, D, causal, seqlens=None, out=None, lse=None):
    """q, k, v: [B*S, H*D] views (own row strides).  Returns (o [B*S, H*D], lse [B,H,S_pad]).  No V re-layout pass."""
    out = torch.empty(B * S, H * D, dtype=q.dtype, device=q.device) if out is None else out
    if lse is None:  # the kernel writes rows < S; the pad rows of a ragged S_pad must read as zeros in the backward
        lse = (torch.empty if S % 64 == 0 else torch.zeros)(B, H, round_up(S, 64), dtype=torch.float32, device=q.device)
    L.check(L.lib().mh_attn_fwd2(p(q), i64(q.stride(0)), p(k), i64(k.stride(0)), p(v), i64(v.stride(0)), p(out), i64(out.stride(0)),
                                 p(lse), p(seqlens), i32(B), i32(S), i32(H), i32(D), i32(int(causal)), i32(dt_of(q)), _stream()), "mh_attn_fwd2")
    return out, lse


_delta_cache = {}


def _delta_ws(B, H, S_pad, device):
    """[2, B, H, S_pad] fp32 scratch of the attention backward, reused across calls (same stream => ordered).  Zeroed once:
    the kernels only ever write rows < S, so the pad rows stay zero."""
    key = (device, B, H, S_pad)
    ws = _delta_cache.get(key)
    if ws is None:
        if len(_delta_cache) > 8:
            _delta_cache.clear()
        ws = _delta_cache[key] = torch.zeros(2, B, H, S_pad, dtype=torch.float32, device=device)
    return ws


_spill_cache = {}
LAST_ATTN_BWD_FORM = None
ATTN_BWD_SPILL = os.environ.get("MH_ATTN_BWD_SPILL", "1") != "0"  # five-product backward (dS spilled by the dK|dV kernel) where it applies


def attn_bwd_spill(on: bool):
    """A/B switch: the causal D = 128 backward runs in its five-product form (dS written by the dK|dV kernel, dQ = dS K as a one-product pass)."""
    global ATTN_BWD_SPILL
    ATTN_BWD_SPILL = bool(on)


ATTN_SPILL_MAX_BYTES = int(float(os.environ.get("MH_ATTN_SPILL_MAX_GB", "12")) * 2 ** 30)  # cap of the dS scratch (cfg 5 needs 8.7 GB)


def release_attn_scratch():
    """Frees the persistent scratch of the attention backward (dS spill of the five-product form, delta / lse rows): call it when a
    process goes from training to inference and wants the HBM back.  The next backward re-allocates what it needs."""
    _spill_cache.clear()
    _delta_cache.clear()


def _spill_ws(B, S, H, device):
    """dS scratch of the five-product attention backward (4.4 GB at cfg 3, 8.7 GB at cfg 5; O(B H S^2)): ONE grow-only buffer per
    (device, stream), shared by every layer (launches on one stream are ordered).  Returns None - the caller then runs the
    seven-product form, which needs no scratch - when the size exceeds ATTN_SPILL_MAX_BYTES (env MH_ATTN_SPILL_MAX_GB, default 12), exceeds a
    quarter of the HBM that is free right now, cannot be allocated (OOM), or a graph capture is in progress.  `release_attn_scratch()` frees it."""
    lib = L.lib()
    lib.mh_attn_bwd_spill_bytes.restype = C.c_int64
    need = int(lib.mh_attn_bwd_spill_bytes(i32(B), i32(S), i32(H)))
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _spill_cache.get(key)
    if ws is None or ws.numel() < need:
        if torch.cuda.is_current_stream_capturing() or need > ATTN_SPILL_MAX_BYTES:
            return None
        have = 0 if ws is None else ws.numel()
        free_b = torch.cuda.mem_get_info(device)[0] + torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
        if need - have > free_b // 4:
            return None
        if len(_spill_cache) > 4:
            _spill_cache.clear()
        _spill_cache.pop(key, None)
        del ws
        try:
            ws = _spill_cache[key] = torch.empty(need, dtype=torch.uint8, device=device)
        except torch.cuda.OutOfMemoryError:
            return None
    return ws


def attn_bwd2(q, k, v, o, do, lse, B, S, H, D, causal, seqlens=None, dq=None, dk=None, dv=None, rope=None, spill=None):
    """Backward without re-layout passes (transpose-read kernels).  rope = (cos, sin) table: dq and dk leave the kernels already rotated
    back (inverse RoPE fused into the epilogues: gradients w.r.t. the UN-rotated q, k).  spill (None = ATTN_BWD_SPILL): the causal D = 128
    case with S % 128 == 0 and no ragged lengths runs as FIVE products - the dK|dV kernel spills dS into a scratch buffer, dQ = dS K."""
    dq = torch.empty(B * S, H * D, dtype=q.dtype, device=q.device) if dq is None else dq
    dk = torch.empty(B * S, H * D, dtype=q.dtype, device=q.device) if dk is None else dk
    dv = torch.empty(B * S, H * D, dtype=q.dtype, device=q.device) if dv is None else dv
    delta = _delta_ws(B, H, round_up(S, 64), q.device)  # [delta | lse*log2e]
    use_spill = ATTN_BWD_SPILL if spill is None else spill
    # (the dS spill lives in attn_bwd3_kv_k = fused-kv mode 2; in modes 0 / 1 the library would run the seven-product kernels and ignore ds_ws)
    use_spill = use_spill and causal and D == 128 and S % 128 == 0 and seqlens is None and int(L.lib().mh_attn_bwd_fused_kv_mode()) == 2
    ws = _spill_ws(B, S, H, q.device) if use_spill else None
    global LAST_ATTN_BWD_FORM
    if causal and D == 128:
        LAST_ATTN_BWD_FORM = "five-product" if ws is not None else "seven-product"  # (tests: which form the last causal D = 128 call took)
    if ws is not None:
        L.check(L.lib().mh_attn_bwd2_spill(p(q), i64(q.stride(0)), p(k), i64(k.stride(0)), p(v), i64(v.stride(0)), p(o), i64(o.stride(0)),
                                           p(do), i64(do.stride(0)), p(lse), p(delta), p(dq), i64(dq.stride(0)), p(dk), i64(dk.stride(0)),
                                           p(dv), i64(dv.stride(0)), p(seqlens), i32(B), i32(S), i32(H), i32(D), i32(int(causal)), p(rope),
                                           i32(dt_of(q)), p(ws), _stream()), "mh_attn_bwd2_spill")
        return dq, dk, dv
    L.check(L.lib().mh_attn_bwd2(p(q), i64(q.stride(0)), p(k), i64(k.stride(0)), p(v), i64(v.stride(0)), p(o), i64(o.stride(0)),
                                 p(do), i64(do.stride(0)), p(lse), p(delta), p(dq), i64(dq.stride(0)), p(dk), i64(dk.stride(0)),
                                 p(dv), i64(dv.stride(0)), p(seqlens), i32(B), i32(S), i32(H), i32(D), i32(int(causal)), p(rope),
                                 i32(dt_of(q)), _stream()), "mh_attn_bwd2")
    return dq, dk, dv


def im2col_patches(pixels, ps, kpad, dtype, rows_per_img=None, row0=0, out=None):
    """pixels [N,3,H,H] -> cols [N*rows_per_img, kpad]; patch p of image n at row n*rows_per_img + row0 + p, rows below
    row0 (the CLS slot) zero.  `out`: write into a caller-provided row range (one launch per image tensor of a batch)."""
    N, Cc, Himg, Wimg = pixels.shape
    assert Cc == 3 and Himg == Wimg and pixels.is_contiguous()
    G = Himg // ps
    rpi = G * G + row0 if rows_per_img is None else rows_per_img
    cols = torch.empty(N * rpi, kpad, dtype=dtype, device=pixels.device) if out is None else out
    assert cols.shape == (N * rpi, kpad) and cols.is_contiguous()
    L.check(L.lib().mh_im2col_patches(p(pixels), i32(dt_of(pixels)), p(cols), i32(N), i32(Himg), i32(ps), i32(kpad), i32(rpi), i32(row0),
                                      i32(dt_of(dtype)), _stream()), "mh_im2col_patches")
    return cols


def copy2d(src, dst, accumulate=False):
    """dst[r, c] (=|+=) src[r, c] over src's [rows, cols] block; both row-major 16-bit with their own row strides."""
    rows, cols = src.shape
    assert dst.shape[0] >= rows and dst.shape[1] >= cols and src.dtype == dst.dtype
    L.check(L.lib().mh_copy2d(p(src), i64(_rowmajor(src)), p(dst), i64(_rowmajor(dst)), i32(rows), i32(cols), i32(int(accumulate)),
                              i32(dt_of(src)), _stream()), "mh_copy2d")
    return dst


def select_tokens(logits, V=None, do_sample=False, temperature=1.0, top_k=0, top_p=1.0, seed=0, step=0, return_u=False):
    """logits fp32 [R, >=V] -> next token ids int64 [R]: greedy argmax, or temperature / top-k / top-p multinomial with a
    counter-based uniform per (seed, step, row) (include/merlin_hip.h: mh_select_tokens)."""
    R = logits.shape[0]
    V = logits.shape[1] if V is None else V
    assert logits.dtype == torch.float32 and logits.stride(1) == 1
    out = torch.empty(R, dtype=torch.int64, device=logits.device)
    u = torch.empty(R, dtype=torch.float32, device=logits.device) if return_u else None
    L.check(L.lib().mh_select_tokens(p(logits), i64(logits.stride(0)), i32(R), i32(V), i32(int(do_sample)), f32(temperature), i32(top_k),
                                     f32(top_p), u64(seed & 0xFFFFFFFFFFFFFFFF), i64(step), p(out), p(u), _stream()), "mh_select_tokens")
    return (out, u) if return_u else out


def log_softmax_rows(logits, V=None, row_bias=None):
    """log_softmax over the first V columns of every row (+ row_bias[r]): fp32 [R, V]."""
    R = logits.shape[0]
    V = logits.shape[1] if V is None else V
    assert logits.dtype == torch.float32 and logits.stride(1) == 1
    out = torch.empty(R, V, dtype=torch.float32, device=logits.device)
    L.check(L.lib().mh_log_softmax_rows(p(logits), i64(logits.stride(0)), i32(R), i32(V), p(out), i64(V), p(row_bias), _stream()), "mh_log_softmax_rows")
    return out


def gather_rows2d(src, idx, dst, cols=None):
    """dst[i, :cols] = src[idx[i], :cols] for row-major 2-D views (own row strides); idx int64 on the device."""
    assert src.dim() == 2 and dst.dim() == 2 and src.stride(1) == 1 and dst.stride(1) == 1 and idx.dtype == torch.int64
    cols = src.shape[1] if cols is None else cols
    es = src.element_size()
    L.check(L.lib().mh_gather_rows2d(p(src), i64(src.stride(0) * es), p(idx), p(dst), i64(dst.stride(0) * es), i32(idx.numel()),
                                     i64(cols * es), _stream()), "mh_gather_rows2d")
    return dst


def gather_rows(table, idx, out=None):
    """out[i, :] = table[idx[i], :] (idx int64 on the device): the embedding-gather kernel without a splice table."""
    assert idx.dtype == torch.int64 and table.is_contiguous()
    return embed_splice_fwd(idx.contiguous().view(-1), None, table, None, out=out)


def vit_assemble(patch, cls, pos, N, G2):
    d = patch.shape[1]
    x = torch.empty(N * (G2 + 1), d, dtype=patch.dtype, device=patch.device)
    L.check(L.lib().mh_vit_assemble(p(patch), p(cls), p(pos), p(x), i32(N), i32(G2), i32(d), i32(dt_of(patch)), _stream()), "mh_vit_assemble")
    return x


def conv3x3_cols(x, N, G, C, stride, rows_per_img, row0):
    Go = (G + 2 - 3) // stride + 1
    cols = torch.empty(N * Go * Go, C * 9, dtype=x.dtype, device=x.device)
    L.check(L.lib().mh_conv3x3_cols(p(x), p(cols), i32(N), i32(G), i32(C), i32(stride), i32(rows_per_img), i32(row0), _stream()), "mh_conv3x3_cols")
    return cols


def conv3x3_col2im(dcols, N, G, C, stride, rows_per_img, row0):
    dx = torch.empty(N * rows_per_img, C, dtype=dcols.dtype, device=dcols.device)
    L.check(L.lib().mh_conv3x3_col2im(p(dcols), p(dx), i32(N), i32(G), i32(C), i32(stride), i32(rows_per_img), i32(row0), i32(dt_of(dcols)), _stream()), "mh_conv3x3_col2im")
    return dx


def splice_index(ids, img_offset, P, im_patch, im_start, im_end, err, rows_per_img=None, row0=0):
    B, S = ids.shape
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    src = torch.empty(B, S, dtype=torch.int32, device=ids.device)
    L.check(L.lib().mh_splice_index(p(ids), p(img_offset), p(src), p(err), i32(B), i32(S), i32(P), i64(im_patch), i64(im_start), i64(im_end), i32(P if rows_per_img is None else rows_per_img), i32(row0), _stream()), "mh_splice_index")
    return src


def mask_lens(mask):
    B, S = mask.shape
    assert mask.is_contiguous() and mask.element_size() == 1
    lens = torch.empty(B, dtype=torch.int32, device=mask.device)
    L.check(L.lib().mh_mask_lens(p(mask), p(lens), i32(B), i32(S), _stream()), "mh_mask_lens")
    return lens


def norm_fwd_f32in(x32, w, eps, b=None, want_x16=False):
    """Reader of the fp32 residual stream: (y, x16 | None) = (norm(x32) in w's 16-bit dtype, 16-bit copy of x32).  b = None: RMSNorm,
    else LayerNorm.  include/merlin_hip.h: mh_norm_fwd_f32in."""
    rows, d = x32.shape
    assert x32.dtype == torch.float32 and x32.is_contiguous() and w.numel() == d
    y = torch.empty(rows, d, dtype=w.dtype, device=x32.device)
    x16 = torch.empty(rows, d, dtype=w.dtype, device=x32.device) if want_x16 else None
    L.check(L.lib().mh_norm_fwd_f32in(p(x32), p(w), p(b), p(y), p(x16), i32(rows), i32(d), f32(eps), i32(dt_of(w)), _stream()), "mh_norm_fwd_f32in")
    return y, x16


def vit_assemble_f32(patch32, cls, pos, N, G2):
    """fp32 form of vit_assemble: patch32 = the patch projection's fp32 output [N * (G2 + 1), d]; returns fp32 x."""
    d = patch32.shape[1]
    assert patch32.dtype == torch.float32 and patch32.is_contiguous()
    x = torch.empty(N * (G2 + 1), d, dtype=torch.float32, device=patch32.device)
    L.check(L.lib().mh_vit_assemble_f32(p(patch32), p(cls), p(pos), p(x), i32(N), i32(G2), i32(d), i32(dt_of(cls)), _stream()), "mh_vit_assemble_f32")
    return x


def layernorm_f32_to_f32(x32, w, b, eps, want_x16=False):
    """(y32, x16 | None) = (LayerNorm(x32) in fp32, 16-bit copy of x32): the CLIP tower's pre_layrnorm at the start of the fp32 stream."""
    rows, d = x32.shape
    assert x32.dtype == torch.float32 and x32.is_contiguous()
    y = torch.empty(rows, d, dtype=torch.float32, device=x32.device)
    x16 = torch.empty(rows, d, dtype=w.dtype, device=x32.device) if want_x16 else None
    L.check(L.lib().mh_layernorm_f32_to_f32(p(x32), p(w), p(b), p(y), p(x16), i32(rows), i32(d), f32(eps), i32(dt_of(w)), _stream()), "mh_layernorm_f32_to_f32")
    return y, x16


def embed_splice_fwd_f32(ids, src, embed, feats32):
    """fp32 form of embed_splice_fwd: embedding rows widened, image rows taken from the projector's fp32 output."""
    T = ids.numel()
    d = embed.shape[1]
    assert feats32 is None or (feats32.dtype == torch.float32 and feats32.is_contiguous())
    out = torch.empty(T, d, dtype=torch.float32, device=embed.device)
    L.check(L.lib().mh_embed_splice_fwd_f32(p(ids), p(src), p(embed), p(feats32), p(out), i32(T), i32(d), i32(dt_of(embed)), _stream()), "mh_embed_splice_fwd_f32")
    return out


def mask_unpad_index(mask):
    """bool/uint8 mask [B, S] -> (fwd int64 [B*S], inv int64 [B*S], count int32 [B]): the unpad / pad row tables of the key-padding
    attention branch (include/merlin_hip.h: mh_mask_unpad_index), consumed by gather_rows2d."""
    B, S = mask.shape
    assert mask.is_contiguous() and mask.element_size() == 1
    fwd = torch.empty(B * S, dtype=torch.int64, device=mask.device)
    inv = torch.empty(B * S, dtype=torch.int64, device=mask.device)
    cnt = torch.empty(B, dtype=torch.int32, device=mask.device)
    L.check(L.lib().mh_mask_unpad_index(p(mask), p(fwd), p(inv), p(cnt), i32(B), i32(S), _stream()), "mh_mask_unpad_index")
    return fwd, inv, cnt


def check_inputs(ids, labels, mask, lens, err, V):
    """Device-side validation (ids / labels in range, right-padded mask, padding present at all); flags land in err[4:11] (int32[12])."""
    ref = ids if ids is not None else (labels if labels is not None else mask)
    B, S = ref.shape
    assert err.numel() >= 12 and err.dtype == torch.int32
    L.check(L.lib().mh_check_inputs(p(ids), p(labels), p(mask), p(lens), p(err), i32(B), i32(S), i32(V), _stream()), "mh_check_inputs")


def embed_splice_fwd(ids, src, embed, feats, out=None):
    T = ids.numel()
    d = embed.shape[1]
    out = torch.empty(T, d, dtype=embed.dtype, device=embed.device) if out is None else out
    L.check(L.lib().mh_embed_splice_fwd(p(ids), p(src), p(embed), p(feats), p(out), i32(T), i32(d), i32(dt_of(embed)), _stream()), "mh_embed_splice_fwd")
    return out


def embed_splice_bwd(ids, src, dout, dfeats, dembed32):
    T, d = dout.shape
    L.check(L.lib().mh_embed_splice_bwd(p(ids), p(src), p(dout), p(dfeats), p(dembed32), i32(T), i32(d), i32(dt_of(dout)), _stream()), "mh_embed_splice_bwd")


def ce_fwd(logits, labels, V):
    """logits fp32 [B*S, ldl]; labels int64 [B, S].  Returns (row_loss[T], lse[T], out2[2])."""
    B, S = labels.shape
    T = B * S
    row_loss = torch.empty(T, dtype=torch.float32, device=logits.device)
    lse = torch.empty(T, dtype=torch.float32, device=logits.device)
    out2 = torch.empty(4, dtype=torch.float32, device=logits.device)
    L.check(L.lib().mh_ce_fwd(p(logits), i64(logits.stride(0)), p(labels), p(row_loss), p(lse), p(out2), i32(B), i32(S), i32(V), _stream()), "mh_ce_fwd")
    return row_loss, lse, out2


def ce_bwd(logits, labels, lse, out2, V, Vpad, gscale, dtype, out=None):
    B, S = labels.shape
    T = B * S
    out = torch.empty(T, Vpad, dtype=dtype, device=logits.device) if out is None else out
    L.check(L.lib().mh_ce_bwd(p(logits), i64(logits.stride(0)), p(labels), p(lse), p(out2), p(out), i64(out.stride(0)), i32(B), i32(S),
                              i32(V), i32(Vpad), f32(gscale), i32(dt_of(dtype)), _stream()), "mh_ce_bwd")
    return out


def ce_bwd_rows(logits, labels, lse, out2, rows, V, Vpad, gscale, dtype):
    """Compact form of ce_bwd: row r of the result is the gradient of logits row rows[r] (int64 flat positions, < 0: a zero row)."""
    S = labels.shape[1]
    n = rows.numel()
    out = torch.empty(n, Vpad, dtype=dtype, device=logits.device)
    L.check(L.lib().mh_ce_bwd_rows(p(logits), i64(logits.stride(0)), p(labels), p(lse), p(out2), p(out), i64(out.stride(0)), p(rows), i32(n), i32(S),
                                   i32(V), i32(Vpad), f32(gscale), i32(dt_of(dtype)), _stream()), "mh_ce_bwd_rows")
    return out


def adamw_(param, grad, m, v, lr, beta1, beta2, eps, wd, step, gscale=1.0):
    L.check(L.lib().mh_adamw(p(param), p(grad), p(m), p(v), i64(param.numel()), f32(lr), f32(beta1), f32(beta2), f32(eps), f32(wd),
                             i32(step), f32(gscale), i32(dt_of(param)), _stream()), "mh_adamw")


def adamw_clip_(param, grad, m, v, lr, beta1, beta2, eps, wd, step, gscale, gscale_dev):
    L.check(L.lib().mh_adamw_clip(p(param), p(grad), p(m), p(v), i64(param.numel()), f32(lr), f32(beta1), f32(beta2), f32(eps), f32(wd),
                                  i32(step), f32(gscale), p(gscale_dev), i32(dt_of(param)), _stream()), "mh_adamw_clip")


def clip_scale(sumsq_t, gscale, max_norm, out2):
    L.check(L.lib().mh_clip_scale(p(sumsq_t), f32(gscale), f32(max_norm), p(out2), _stream()), "mh_clip_scale")


def sumsq_det(g, partial, out):
    L.check(L.lib().mh_sumsq_det(p(g), i64(g.numel()), p(partial), p(out), i32(dt_of(g)), _stream()), "mh_sumsq_det")


def any_nonzero(g, flag):
    """flag (int32[1], zeroed by the caller) |= 1 when any 16-bit element of g is non-zero (exact, on the bit patterns)."""
    L.check(L.lib().mh_any_nonzero(p(g), i64(g.numel()), p(flag), _stream()), "mh_any_nonzero")


def sumsq(g, out):
    L.check(L.lib().mh_sumsq(p(g), i64(g.numel()), p(out), i32(dt_of(g)), _stream()), "mh_sumsq")


_gemv_mfma_min = 3


def gemv_mfma_min_rows(rows: int):
    """A/B switch: row count from which gemv / gemv_fp8w use the MFMA kernel (<= 0 restores the measured default, 3; 17 = never)."""
    global _gemv_mfma_min
    _gemv_mfma_min = rows if rows > 0 else 3
    L.lib().mh_gemv_mfma_min_rows(i32(rows))


def gemv_mfma_pair_min_rows(rows16: int, rows_fp8: int):
    """A/B switch: row counts from which the SwiGLU / RoPE-append projections use the MFMA form (<= 0: defaults 6 / 4)."""
    L.lib().mh_gemv_mfma_pair_min_rows(i32(rows16), i32(rows_fp8))


def _gemv_rows_per_launch(K):
    """Activation rows one mh_gemv / mh_gemv_fp8w launch takes: 16 through the MFMA form (needs K % 32 == 0 and the MFMA row
    threshold within reach), 8 through the wave-per-row form otherwise (csrc/decode.hip gemv_impl returns MH_ERR_ARG for 9-16
    rows there: e.g. beam search with 9-16 beam rows on a model whose K is a multiple of 8 but not of 32)."""
    return 16 if (K % 32 == 0 and _gemv_mfma_min <= 9) else 8


def _gemv_fused_rows_ok(M, K):
    """Rows the fused decode projections (SwiGLU / RoPE epilogues) take in one launch: 8 as one wave per row pair, 16 in the MFMA form."""
    return M <= 8 or (M <= 16 and M >= _gemv_mfma_min and K % 64 == 0)


def gemv_ksplit(on: bool):
    """A/B switch: K split over a block's waves in the 1-2 row GEMV at N <= 8192 (default on)."""
    L.lib().mh_gemv_ksplit(i32(1 if on else 0))


def gemv_mfma_wide(on: bool):
    """A/B switch: 16 instead of 8 waves per block in the MFMA GEMV at N <= 8192 (default on)."""
    L.lib().mh_gemv_mfma_wide(i32(1 if on else 0))


def attn_fwd_pingpong(on):
    """A/B switch of the D = 128 attention forward: 0 / False = attn_fwd2 (default), 1 / True = the ping-pong form (8-wave blocks, SIMD partners in
    opposite phases; csrc/attn_fwd3.hip), 2 = one wave per SIMD with 64 query rows per wave (csrc/attn_fwd4.hip)."""
    L.lib().mh_attn_fwd_pingpong(i32(int(on)))


def attn_bwd_fused_kv(on):
    """A/B switch for dK and dV of the D = 128 attention backward: True / 2 = attn_bwd3_kv_k (default: one kernel, register-staged copies,
    continuous fragment stream), 1 = attn_bwd2_kv_k<MODE 3> (one kernel, LDS-DMA copies: rounds 2-4), False / 0 = two kernels."""
    L.lib().mh_attn_bwd_fused_kv(i32(2 if on is True else int(on)))


def gemm_raster_group(gm: int):
    """A/B switch: tile rows per raster group of the MFMA GEMM kernels (default 4)."""
    L.lib().mh_gemm_raster_group(i32(gm))


def gemm_persistent(on: bool):
    """A/B switch: persistent launch of the 256-tile GEMM kernels (default on; env MH_GEMM_PERSISTENT=0 turns it off at import)."""
    L.lib().mh_gemm_persistent(i32(1 if on else 0))


def gemm_w4_policy(mask: int):
    """A/B switch: operand layouts the auto selection sends to the 4-wave 256x256 GEMM (bit 0 TN, bit 1 NN, bit 2 NT, bit 3 fp8 NT; default 11)."""
    L.lib().mh_gemm_w4_policy(i32(mask))


def gemm_w4_half(mode: int):
    """128-row block tiles of the 4-wave GEMM for NT products with few rows (short prefills): 0 = never, 1 = auto (default), 2 = wherever the form exists."""
    L.lib().mh_gemm_w4_half(i32(mode))


def gemm_force_kernel(which: int):
    """0 = auto, 128 / 256 = force that tile size (tests, A/B benchmarks).  Other codes select the development arms and
    exist only in the dev library (tools/dev_arms/)."""
    L.lib().mh_gemm_force_kernel(i32(which))
