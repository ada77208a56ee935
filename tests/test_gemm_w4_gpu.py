"""csrc/gemm_w4.hip, round-4 forms: the fused forward epilogues on the fp32 accumulators (q|k|v + RoPE, gate|up + SwiGLU), the
SwiGLU-backward dgrad, fp32 outputs (plain / accumulating), split-K with fp32 partials, K tails of K-strided operands (rows >= K read as
zeros through the buffer descriptor's range check) and the grouped weight-gradient launch - each against a plain fp32 torch reference of
the same op on 16-bit-rounded inputs, and against the 8-wave kernel where that kernel has the same form."""
import pytest
import torch

from test_ops_gpu import DTYPES, EPS16, dev, relerr, rnd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from merlin_amd import ops as O

    assert O.arch_ok(0), "not a gfx950 device"
    return O


def _with_kernel(ops, which, fn):
    ops.gemm_force_kernel(which)
    try:
        return fn()
    finally:
        ops.gemm_force_kernel(0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,S,H,K", [(613, 613, 4, 256), (1024, 512, 6, 384), (300, 100, 2, 128)])
def test_w4_qkv_projection_with_rope_on_the_accumulators(ops, dtype, T, S, H, K):
    D = 128
    N = 3 * H * D
    a, w = rnd(T, K, dtype=dtype), rnd(N, K, dtype=dtype, seed=1, scale=0.3)
    tab = ops.rope_table(S, D, 10000.0, dev())
    ref = (a.float() @ w.float().t()).view(T, 3, H, D)
    pos = torch.arange(T, device=dev()) % S
    cos, sin = tab[pos, :, 0][:, None, :], tab[pos, :, 1][:, None, :]  # [T, 1, 64]
    want = ref.clone()
    for part in (0, 1):
        x = ref[:, part]
        lo, hi = x[..., :64], x[..., 64:]
        want[:, part] = torch.cat([lo * cos - hi * sin, hi * cos + lo * sin], -1)
    want = want.view(T, N)
    got4 = _with_kernel(ops, 4, lambda: ops.gemm_nt_rope(a, w, tab, S, H, D))
    got8 = _with_kernel(ops, 256, lambda: ops.gemm_nt_rope(a, w, tab, S, H, D))
    assert relerr(got4, want) < 3 * EPS16[dtype]
    assert relerr(got4, got8.float()) < 3 * EPS16[dtype]
    # the fp32-accumulator form rounds once, the staged form twice (GEMM output, then the rotated value): it is never further from fp32
    e4, e8 = float((got4.float() - want).pow(2).mean().sqrt()), float((got8.float() - want).pow(2).mean().sqrt())
    assert e4 <= e8 * 1.02, (e4, e8)
    assert torch.equal(got4, _with_kernel(ops, 4, lambda: ops.gemm_nt_rope(a, w, tab, S, H, D)))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,ff,K", [(613, 1408, 256), (1024, 11008, 128), (260, 640, 384), (256, 200, 128)])
def test_w4_gate_up_projection_with_swiglu_on_the_accumulators(ops, dtype, T, ff, K):
    x, wgu = rnd(T, K, dtype=dtype), rnd(2 * ff, K, dtype=dtype, seed=1, scale=0.3)
    gu_ref = x.float() @ wgu.float().t()
    act_ref = torch.nn.functional.silu(gu_ref[:, :ff]) * gu_ref[:, ff:]
    gu4, act4 = _with_kernel(ops, 4, lambda: ops.gemm_swiglu_fwd(x, wgu))
    gu8, act8 = _with_kernel(ops, 256, lambda: ops.gemm_swiglu_fwd(x, wgu))
    assert relerr(gu4, gu_ref) < 3 * EPS16[dtype] and relerr(act4, act_ref) < 3 * EPS16[dtype]
    assert relerr(gu4, gu8.float()) < 2 * EPS16[dtype]  # the stored gate|up values: one rounding of (nearly) the same accumulators
    e4, e8 = float((act4.float() - act_ref).pow(2).mean().sqrt()), float((act8.float() - act_ref).pow(2).mean().sqrt())
    assert e4 <= e8 * 1.02, (e4, e8)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,ff,d", [(613, 1408, 256), (512, 11008, 128), (300, 640, 384)])
def test_w4_down_dgrad_with_swiglu_backward(ops, dtype, T, ff, d):
    dy, wd, gu = rnd(T, d, dtype=dtype), rnd(d, ff, dtype=dtype, seed=1, scale=0.3), rnd(T, 2 * ff, dtype=dtype, seed=2)
    g32, u32 = gu[:, :ff].float().requires_grad_(True), gu[:, ff:].float().requires_grad_(True)
    dact = dy.float() @ wd.float()
    (torch.nn.functional.silu(g32) * u32).backward(dact)
    want = torch.cat([g32.grad, u32.grad], 1)
    got4 = _with_kernel(ops, 4, lambda: ops.gemm_swiglu_bwd(dy, wd, gu))
    got8 = _with_kernel(ops, 256, lambda: ops.gemm_swiglu_bwd(dy, wd, gu))
    assert relerr(got4, want) < 4 * EPS16[dtype]
    assert relerr(got4, got8.float()) < 3 * EPS16[dtype]  # same products, dact rounded to 16 bits in both, same element maths


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(613, 4096, 256), (1000, 32064, 128), (256, 520, 384)])
def test_w4_fp32_outputs_plain_and_accumulating(ops, dtype, M, N, K):
    a, b = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, seed=1, scale=0.3)
    ref = a.float() @ b.float().t()
    out4 = _with_kernel(ops, 4, lambda: ops.gemm_nt(a, b, out_f32=True))
    assert relerr(out4, ref) < 1e-5
    stream = torch.randn(M, N, device=dev())
    s4, s8 = stream.clone(), stream.clone()
    _with_kernel(ops, 4, lambda: ops.gemm_nt(a, b, out=s4, accum=True))
    _with_kernel(ops, 256, lambda: ops.gemm_nt(a, b, out=s8, accum=True))
    assert relerr(s4, stream + ref) < 1e-5 and relerr(s4, s8) < 1e-6


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,M,N,splits", [(27696, 1024, 1024, 4), (4096, 512, 768, 3), (1000, 264, 256, 2), (8192, 256, 4096, 1)])
def test_w4_weight_gradient_k_tail_and_split_k(ops, dtype, T, M, N, splits):
    """TN products over token counts that are not a multiple of the 128-row K-tile pair (the CLIP tower's 48 x 577 = 27 696 tokens):
    rows >= T arrive in LDS as zeros.  Direct 16-bit store and the split-K form (fp32 partials + the fixed-order reduce pass)."""
    from merlin_amd import ops as O

    dy, x = rnd(T, M, dtype=dtype, scale=0.5), rnd(T, N, dtype=dtype, seed=1, scale=0.5)
    # poison behind the operands: the kernel must not read rows >= T (NaNs would surface in every output)
    ref = dy.float().t() @ x.float()

    def call(which):
        out = torch.zeros(M, N, dtype=dtype, device=dev())
        ws = torch.full((max(1, splits) * M * N,), float("nan"), dtype=torch.float32, device=dev()) if splits > 1 else None

        def go():
            O.L.check(O.L.lib().mh_gemm_splitk(O.p(dy), O.i64(M), O.i32(1), O.p(x), O.i64(N), O.i32(1), O.p(out), O.i64(N), O.i32(M), O.i32(N), O.i32(T),
                                               O.i32(O.dt_of(dy)), O.i32(0), O.i32(0), O.i32(splits), O.p(ws), O._stream()), "mh_gemm_splitk")
            return out
        return _with_kernel(ops, which, go)

    got4, got8 = call(4), call(256)
    assert relerr(got4, ref) < 3 * EPS16[dtype], "4-wave"
    assert relerr(got8, ref) < 3 * EPS16[dtype], "8-wave"
    assert relerr(got4, got8.float()) < 2 * EPS16[dtype]
    assert torch.equal(got4, call(4))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,vd,vff", [(27696, 1024, 4096), (1154, 128, 256), (2048, 264, 520)])
def test_grouped_weight_gradients_of_a_clip_layer(ops, dtype, T, vd, vff):
    """mh_wgrad_grouped on the four Linears of a CLIP encoder layer (fc2, fc1, out_proj, q|k|v) over T tokens: every output vs fp32 torch
    (sampled tiles at the real geometry) and vs the per-problem launches; fresh and accumulating; deterministic."""
    shapes = [(vd, vff), (vff, vd), (vd, vd), (3 * vd, vd)]
    probs, refs = [], []
    for i, (M, N) in enumerate(shapes):
        dy, x = rnd(T, M, dtype=dtype, seed=2 * i, scale=0.5), rnd(T, N, dtype=dtype, seed=2 * i + 1, scale=0.5)
        probs.append((dy, x, torch.zeros(M, N, dtype=dtype, device=dev())))
    ops.wgrad_tn_grouped(probs, accum=False, force=True)
    singles = []
    for dy, x, out in probs:
        o1 = torch.zeros_like(out)
        ops.wgrad_tn(dy, x, o1, accum=False)
        singles.append(o1)
    for (dy, x, out), o1 in zip(probs, singles):
        M, N = out.shape
        for r0, c0 in ((0, 0), (max(0, M - 256), max(0, N - 256)), (M // 2 // 8 * 8, 0)):
            r1, c1 = min(M, r0 + 256), min(N, c0 + 256)
            ref = dy[:, r0:r1].float().t() @ x[:, c0:c1].float()
            assert relerr(out[r0:r1, c0:c1], ref) < 3 * EPS16[dtype], (M, N, r0, c0)
        assert relerr(out, o1.float()) < 2 * EPS16[dtype]
    again = [(dy, x, torch.zeros_like(out)) for dy, x, out in probs]
    ops.wgrad_tn_grouped(again, accum=False, force=True)
    assert all(torch.equal(a[2], b[2]) for a, b in zip(again, probs))
    # accumulating form
    olds = [rnd(*out.shape, dtype=dtype, seed=50 + i) for i, (_, _, out) in enumerate(probs)]
    acc = [(dy, x, old.clone()) for (dy, x, _), old in zip(probs, olds)]
    ops.wgrad_tn_grouped(acc, accum=True, force=True)
    for (_, _, got), (_, _, fresh), old in zip(acc, probs, olds):
        assert relerr(got, fresh.float() + old.float()) < 3 * EPS16[dtype]


# ---- fp8 operands: the fused forms of the fp8 training step on the 4-wave fp8 kernel (gemm_w4_f8<EK>) -------------------------------------------
def _deq(q8):
    q, s = q8[0], q8[1]
    return q.view(torch.float8_e4m3fn).float() * s[:, None]


@pytest.mark.parametrize("dtype", DTYPES)
def test_w4_fp8_fused_forms_vs_dequantised_fp32_and_the_8wave_kernel(ops, dtype):
    """q|k|v + RoPE, gate|up + SwiGLU, the SwiGLU-backward dgrad and the fp32 logits store with e4m3 operands (per-row scales, exponent-free weights):
    against the fp32 product of the DEQUANTISED operands followed by the fp32 element-wise op, and against the 8-wave fp8 kernel's staged forms."""
    T, K, S, H, D, ff = 600, 512, 300, 2, 128, 640
    x = rnd(T, K, dtype=dtype)
    a8 = ops.quant_fp8_rows(x)
    xa = _deq(a8)
    tab = ops.rope_table(S, D, 10000.0, dev())
    # q|k|v + RoPE
    wq = rnd(3 * H * D, K, dtype=dtype, seed=1, scale=0.3)
    w8 = ops.quant_fp8_rows(wq)
    ref = (xa @ _deq(w8).t()).view(T, 3, H, D)
    pos = torch.arange(T, device=dev()) % S
    cos, sin = tab[pos, :, 0][:, None, :], tab[pos, :, 1][:, None, :]
    want = ref.clone()
    for part in (0, 1):
        lo, hi = ref[:, part, :, :64], ref[:, part, :, 64:]
        want[:, part] = torch.cat([lo * cos - hi * sin, hi * cos + lo * sin], -1)
    want = want.view(T, -1)
    g4 = _with_kernel(ops, 4, lambda: ops.gemm_fp8_rope(a8, w8, tab, S, H, D, out_dtype=dtype))
    g8 = _with_kernel(ops, 256, lambda: ops.gemm_fp8_rope(a8, w8, tab, S, H, D, out_dtype=dtype))
    assert relerr(g4, want) < 3 * EPS16[dtype] and relerr(g4, g8.float()) < 3 * EPS16[dtype]
    # gate|up + SwiGLU
    wgu = rnd(2 * ff, K, dtype=dtype, seed=2, scale=0.3)
    wgu8 = ops.quant_fp8_rows(wgu)
    gu_ref = xa @ _deq(wgu8).t()
    act_ref = torch.nn.functional.silu(gu_ref[:, :ff]) * gu_ref[:, ff:]
    gu4, act4 = _with_kernel(ops, 4, lambda: ops.gemm_fp8_swiglu_fwd(a8, wgu8, out_dtype=dtype))
    gu8, act8 = _with_kernel(ops, 256, lambda: ops.gemm_fp8_swiglu_fwd(a8, wgu8, out_dtype=dtype))
    assert relerr(gu4, gu_ref) < 3 * EPS16[dtype] and relerr(act4, act_ref) < 3 * EPS16[dtype]
    assert relerr(gu4, gu8.float()) < 2 * EPS16[dtype] and relerr(act4, act8.float()) < 3 * EPS16[dtype]
    # SwiGLU-backward dgrad: dy [T, d] x Wd^T [ff, d]
    d = K
    dy, wdT = rnd(T, d, dtype=dtype, seed=3, scale=0.5), rnd(ff, d, dtype=dtype, seed=4, scale=0.3)
    dy8, wdT8 = ops.quant_fp8_rows(dy), ops.quant_fp8_rows(wdT)
    gu = rnd(T, 2 * ff, dtype=dtype, seed=5)
    g32, u32 = gu[:, :ff].float().requires_grad_(True), gu[:, ff:].float().requires_grad_(True)
    (torch.nn.functional.silu(g32) * u32).backward(_deq(dy8) @ _deq(wdT8).t())
    wantb = torch.cat([g32.grad, u32.grad], 1)
    b4 = _with_kernel(ops, 4, lambda: ops.gemm_fp8_swiglu_bwd(dy8, wdT8, gu))
    b8 = _with_kernel(ops, 256, lambda: ops.gemm_fp8_swiglu_bwd(dy8, wdT8, gu))
    assert relerr(b4, wantb) < 4 * EPS16[dtype] and relerr(b4, b8.float()) < 3 * EPS16[dtype]
    # fp32 logits store
    wl = rnd(1032, K, dtype=dtype, seed=6, scale=0.3)
    wl8 = ops.quant_fp8_rows(wl)
    lg4 = _with_kernel(ops, 4, lambda: ops.gemm_fp8(a8, wl8, out=torch.empty(T, 1032, dtype=torch.float32, device=dev()), dt16=dtype))
    assert relerr(lg4, xa @ _deq(wl8).t()) < 1e-4  # (fp32 store; the reference's own fp32 matmul order and the scale products differ by ~2e-5)
    # plain + residual on the shared store phase
    resid = rnd(T, 1032, dtype=dtype, seed=7)
    r4 = _with_kernel(ops, 4, lambda: ops.gemm_fp8(a8, wl8, out_dtype=dtype, resid=resid))
    assert relerr(r4, xa @ _deq(wl8).t() + resid.float()) < 3 * EPS16[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [1, 64, 101, 128, 129, 613, 700])
def test_w4_half_tiles_are_bit_identical_to_the_256_row_tiles(ops, dtype, M):
    """gemm_w4<..., MI = 4>: the 128-row block tile for products with few rows (a 613-token prefill: q|k|v 144 -> 240 tiles, gate|up 258 -> 430
    half tiles = 1.7 half-rounds instead of two rounds).  An output element is accumulated over the same K-tiles in the same order by the same
    MFMA and goes through the same store phase, so every NT form must equal the 256-row kernel BIT FOR BIT: plain / residual / accumulating
    16-bit stores, fp32 store and accumulate, q|k|v + RoPE, gate|up + SwiGLU."""
    K, H, D, ff, S = 384, 3, 128, 712, 97
    a = rnd(M, K, dtype=dtype)
    w, wqkv, wgu = rnd(520, K, dtype=dtype, seed=1, scale=0.3), rnd(3 * H * D, K, dtype=dtype, seed=2, scale=0.3), rnd(2 * ff, K, dtype=dtype, seed=3, scale=0.3)
    res = rnd(M, 520, dtype=dtype, seed=4)
    old16, old32 = rnd(M, 520, dtype=dtype, seed=5), torch.randn(M, 520, device=dev())
    tab = ops.rope_table(S, D, 10000.0, dev())

    def run(mode):
        ops.gemm_w4_half(mode)
        try:
            def go():
                o16, o32 = old16.clone(), old32.clone()
                ops.gemm_nt(a, w, out=o16, accum=True)
                ops.gemm_nt(a, w, out=o32, accum=True)
                return (ops.gemm_nt(a, w), ops.gemm_nt(a, w, resid=res), o16, ops.gemm_nt(a, w, out_f32=True), o32,
                        ops.gemm_nt_rope(a, wqkv, tab, S, H, D), *ops.gemm_swiglu_fwd(a, wgu))
            return _with_kernel(ops, 4, go)
        finally:
            ops.gemm_w4_half(1)

    full, half = run(0), run(2)
    names = ["plain", "residual", "accumulate16", "fp32", "accumulate32", "rope", "swiglu gu", "swiglu act"]
    for nm, f, h in zip(names, full, half):
        assert torch.isfinite(h.float()).all(), nm
        assert torch.equal(f, h), (nm, float((f.float() - h.float()).abs().max()))
    assert relerr(half[0], a.float() @ w.float().t()) < 3 * EPS16[dtype]
