"""Per-kernel numerics on a real MI355X: every C-ABI op vs a plain PyTorch fp32 reference of the
same op, on inputs rounded to the 16-bit storage type (so the comparison measures the kernel, not
the input quantisation).  Tolerances are relative to the reference's max magnitude."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DTYPES = [torch.bfloat16, torch.float16]
EPS16 = {torch.bfloat16: 2 ** -8, torch.float16: 2 ** -11}


def dev():
    return torch.device("cuda:0")


def rnd(*shape, dtype=torch.bfloat16, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev())


def relerr(got, ref):
    got, ref = got.float(), ref.float()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-20))


@pytest.fixture(scope="module")
def ops():
    from merlin_amd import ops as O

    assert O.arch_ok(0), "not a gfx950 device"
    return O


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (613, 4096, 1024), (200, 103, 256), (577, 3072, 1024), (1000, 11008, 256), (64, 384, 640)])
def test_gemm_plain(ops, dtype, M, N, K):
    a, b = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, seed=1)
    ref = a.float() @ b.float().t()
    out = ops.gemm_nt(a, b)
    assert out.shape == (M, N)
    assert relerr(out, ref) < 3 * EPS16[dtype]
    out32 = ops.gemm_nt(a, b, out_f32=True)
    assert relerr(out32, ref) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_asymmetric_identity(ops, dtype):
    """A = I with an asymmetric B catches a transposed C write (cdna guide rule 16)."""
    K = 128
    a = torch.eye(K, dtype=dtype, device=dev())
    b = (torch.arange(K * K, device=dev()).reshape(K, K) % 251).to(dtype)  # b[n, k]
    out = ops.gemm_nt(a, b)  # out[m, n] = b[n, m]
    assert torch.equal(out.float(), b.float().t())


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_epilogues(ops, dtype):
    M, N, K = 300, 512, 192
    a, b = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, seed=1, scale=0.1)
    bias, resid = rnd(N, dtype=dtype, seed=2), rnd(M, N, dtype=dtype, seed=3)
    base = a.float() @ b.float().t()
    tol = 4 * EPS16[dtype]
    assert relerr(ops.gemm_nt(a, b, bias=bias), base + bias.float()) < tol
    z = base + bias.float()
    assert relerr(ops.gemm_nt(a, b, bias=bias, act="quick_gelu"), z * torch.sigmoid(1.702 * z)) < tol
    assert relerr(ops.gemm_nt(a, b, bias=bias, resid=resid), z + resid.float()) < tol
    acc = rnd(M, N, dtype=dtype, seed=4)
    ref = acc.float() + base
    ops.gemm_nt(a, b, out=acc, accum=True)
    assert relerr(acc, ref) < tol
    acc32 = torch.ones(M, N, device=dev())
    ops.gemm_nt(a, b, out=acc32, accum=True)
    assert relerr(acc32, base + 1) < 1e-5
    # odd N / odd ldc scalar path
    b2 = rnd(103, K, dtype=dtype, seed=5)
    out = torch.zeros(M, 103, dtype=torch.float32, device=dev())
    ops.gemm_nt(a, b2, out=out)
    assert relerr(out, a.float() @ b2.float().t()) < 1e-5
    # strided views (fused qkv style): A is a column slice, C a column slice
    big = rnd(M, 3 * K, dtype=dtype, seed=6)
    cbig = torch.zeros(M, 2 * N, dtype=dtype, device=dev())
    ops.gemm_nt(big[:, K:2 * K], b, out=cbig[:, N:])
    assert relerr(cbig[:, N:], big[:, K:2 * K].float() @ b.float().t()) < tol
    assert float(cbig[:, :N].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
def test_transpose(ops, dtype):
    x = rnd(613, 200, dtype=dtype)
    out = ops.transpose16(x, r_pad=640)
    assert out.shape == (200, 640)
    assert torch.equal(out[:, :613], x.t())
    assert float(out[:, 613:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,d", [(613, 4096), (37, 256), (1000, 1024), (5, 128)])
def test_rmsnorm(ops, dtype, rows, d):
    x, w = rnd(rows, d, dtype=dtype), (1 + 0.1 * rnd(d, dtype=dtype, seed=1).float()).to(dtype)
    eps = 1e-6

    def f(x32, w32):
        return w32 * x32 * torch.rsqrt(x32.pow(2).mean(-1, keepdim=True) + eps)

    y = ops.rmsnorm_fwd(x, w, eps)
    assert relerr(y, f(x.float(), w.float())) < 2 * EPS16[dtype]
    dy = rnd(rows, d, dtype=dtype, seed=2)
    x32, w32 = x.float().requires_grad_(), w.float().requires_grad_()
    f(x32, w32).backward(dy.float())
    dw = torch.zeros(d, dtype=torch.float32, device=dev())
    dx = ops.rmsnorm_bwd(x, w, dy, eps, dw_out=dw)
    assert relerr(dx, x32.grad) < 3 * EPS16[dtype]
    assert relerr(dw, w32.grad) < 1e-4
    # accumulate paths
    dx0 = rnd(rows, d, dtype=dtype, seed=3)
    ref = dx0.float() + x32.grad
    ops.rmsnorm_bwd(x, w, dy, eps, dx=dx0, accumulate_dx=True, dw_out=dw, dw_accumulate=True)
    assert relerr(dx0, ref) < 3 * EPS16[dtype]
    assert relerr(dw, 2 * w32.grad) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,d", [(577 * 2, 1024), (34, 128)])
def test_layernorm(ops, dtype, rows, d):
    x = rnd(rows, d, dtype=dtype) + 0.5
    w, b = (1 + 0.1 * rnd(d, dtype=dtype, seed=1).float()).to(dtype), rnd(d, dtype=dtype, seed=2, scale=0.1)
    eps = 1e-5
    y = ops.layernorm_fwd(x, w, b, eps)
    x32, w32, b32 = x.float().requires_grad_(), w.float().requires_grad_(), b.float().requires_grad_()
    ref = torch.nn.functional.layer_norm(x32, (d,), w32, b32, eps)
    assert relerr(y, ref) < 2 * EPS16[dtype]
    dy = rnd(rows, d, dtype=dtype, seed=3)
    ref.backward(dy.float())
    dw = torch.zeros(d, dtype=torch.float32, device=dev())
    db = torch.zeros(d, dtype=torch.float32, device=dev())
    dx = ops.layernorm_bwd(x, w, dy, eps, dw_out=dw, db_out=db)
    assert relerr(dx, x32.grad) < 3 * EPS16[dtype]
    assert relerr(dw, w32.grad) < 1e-4
    assert relerr(db, b32.grad) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
def test_elementwise(ops, dtype):
    rows, ff = 77, 512
    gu = rnd(rows, 2 * ff, dtype=dtype)
    g32, u32 = gu[:, :ff].float().requires_grad_(), gu[:, ff:].float().requires_grad_()
    ref = torch.nn.functional.silu(g32) * u32
    assert relerr(ops.swiglu_fwd(gu), ref) < 2 * EPS16[dtype]
    dout = rnd(rows, ff, dtype=dtype, seed=1)
    ref.backward(dout.float())
    dgu = ops.swiglu_bwd(gu, dout)
    assert relerr(dgu[:, :ff], g32.grad) < 3 * EPS16[dtype]
    assert relerr(dgu[:, ff:], u32.grad) < 3 * EPS16[dtype]
    x = rnd(rows, ff, dtype=dtype, seed=2)
    x32 = x.float().requires_grad_()
    r = x32 * torch.sigmoid(1.702 * x32)
    assert relerr(ops.quick_gelu_fwd(x), r) < 2 * EPS16[dtype]
    r.backward(dout.float())
    assert relerr(ops.quick_gelu_bwd(x, dout), x32.grad) < 3 * EPS16[dtype]
    assert relerr(ops.add(x, dout), x.float() + dout.float()) < 2 * EPS16[dtype]
    s = torch.zeros(1, device=dev())
    ops.sumsq(x, s)
    assert abs(float(s) - float(x.float().pow(2).sum())) < 1e-3 * float(x.float().pow(2).sum())
    y32 = torch.empty(rows, ff, device=dev())
    ops.convert(x, y32)
    assert torch.equal(y32, x.float())


def test_fill_normal_matches_numpy(ops):
    from merlin_amd import weights as W

    name = "model.layers.3.mlp.up_proj.weight"
    ref = W.generate(name, (1000, 96))
    for dtype in (torch.float32, torch.bfloat16, torch.float16):
        t = torch.empty(1000, 96, dtype=dtype, device=dev())
        ops.fill_normal_(t, W.param_key(name, 0))
        assert np.array_equal(t.float().cpu().numpy(), ref), dtype
    nref = W.generate("model.norm.weight", (512,))
    t = torch.empty(512, dtype=torch.bfloat16, device=dev())
    ops.fill_normal_(t, W.param_key("model.norm.weight", 0), offset=1.0)
    assert np.array_equal(t.float().cpu().numpy(), nref)
    # start offset = slicing
    t2 = torch.empty(96 * 10, dtype=torch.float32, device=dev())
    ops.fill_normal_(t2, W.param_key(name, 0), start=96 * 5)
    assert np.array_equal(t2.cpu().numpy(), ref[5:15].reshape(-1))


def _rope_ref(x, S, theta):
    # x [B, S, H, D] fp32; HF rotate-half
    D = x.shape[-1]
    inv = 1.0 / (theta ** (torch.arange(0, D, 2, dtype=torch.float32, device=x.device) / D))
    fr = torch.outer(torch.arange(S, dtype=torch.float32, device=x.device), inv)
    emb = torch.cat((fr, fr), -1)
    cos, sin = emb.cos()[None, :, None, :], emb.sin()[None, :, None, :]
    h = D // 2
    rot = torch.cat((-x[..., h:], x[..., :h]), -1)
    return x * cos + rot * sin


@pytest.mark.parametrize("dtype", DTYPES)
def test_rope(ops, dtype):
    B, S, H, D = 2, 75, 2, 128
    qkv = rnd(B * S, 3 * H * D, dtype=dtype)
    tab = ops.rope_table(S, D, 10000.0, dev())
    ref = qkv.float().view(B, S, 3, H, D).clone()
    ref[:, :, 0] = _rope_ref(ref[:, :, 0], S, 10000.0)
    ref[:, :, 1] = _rope_ref(ref[:, :, 1], S, 10000.0)
    orig = qkv.clone()
    ops.rope_qk_(qkv, tab, S, H, D)
    assert relerr(qkv.view(B, S, 3, H, D), ref) < 3 * EPS16[dtype]
    assert torch.equal(qkv.view(B, S, 3, H, D)[:, :, 2], orig.view(B, S, 3, H, D)[:, :, 2])
    ops.rope_qk_(qkv, tab, S, H, D, inverse=True)  # R^T R = I
    assert relerr(qkv, orig) < 4 * EPS16[dtype]


def _attn_ref(q, k, v, causal, lens):
    # q,k,v [B,S,H,D] fp32 -> o [B,S,H,D], zero rows at padded queries
    B, S, H, D = q.shape
    qh, kh, vh = (t.permute(0, 2, 1, 3) for t in (q, k, v))
    s = (qh @ kh.transpose(-1, -2)) / math.sqrt(D)
    mask = torch.zeros(B, 1, S, S, dtype=torch.bool, device=q.device)
    if causal:
        mask |= torch.ones(S, S, dtype=torch.bool, device=q.device).triu(1)
    ar = torch.arange(S, device=q.device)
    keypad = ar[None, :] >= lens[:, None]
    mask = mask | keypad[:, None, None, :]
    s = s.masked_fill(mask, float("-inf"))
    p = torch.softmax(s, -1)
    p = torch.nan_to_num(p, nan=0.0)
    o = (p @ vh).permute(0, 2, 1, 3)
    qpad = (ar[None, :] >= lens[:, None])[:, :, None, None]
    return o.masked_fill(qpad, 0.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,S,H,D,causal,lens", [
    (1, 128, 1, 128, True, None), (2, 613, 2, 128, True, None), (2, 300, 2, 128, True, [300, 177]),
    (3, 577, 2, 64, False, None), (1, 17, 2, 64, False, None), (2, 40, 2, 128, True, [29, 40]), (1, 1024, 1, 128, True, None)])
def test_attention_fwd_bwd(ops, dtype, B, S, H, D, causal, lens):
    qkv = rnd(B * S, 3 * H * D, dtype=dtype, scale=1.0)
    q, k, v = (qkv[:, i * H * D:(i + 1) * H * D] for i in range(3))
    lens_t = torch.tensor(lens if lens else [S] * B, dtype=torch.int32, device=dev())
    o, lse = ops.attn_fwd2(q, k, v, B, S, H, D, causal, seqlens=lens_t if lens else None)
    q32, k32, v32 = (t.float().reshape(B, S, H, D).clone().requires_grad_() for t in (q, k, v))
    ref = _attn_ref(q32, k32, v32, causal, lens_t.long())
    assert relerr(o.view(B, S, H, D), ref) < 4 * EPS16[dtype], "forward"
    # lse check on valid rows
    sc = (q32.permute(0, 2, 1, 3) @ k32.permute(0, 2, 3, 1)) / math.sqrt(D)
    ar = torch.arange(S, device=dev())
    m = (ar[None, :] >= lens_t.long()[:, None])[:, None, None, :]
    if causal:
        m = m | torch.ones(S, S, dtype=torch.bool, device=dev()).triu(1)
    lse_ref = torch.logsumexp(sc.masked_fill(m, float("-inf")), -1)
    for b in range(B):
        n = int(lens_t[b])
        assert float((lse[b, :, :n] - lse_ref[b, :, :n].detach()).abs().max()) < 2e-2
    do = rnd(B * S, H * D, dtype=dtype, seed=9)
    do_masked = do.clone()
    ref.backward(do.float().view(B, S, H, D))
    tol = 8 * EPS16[dtype]
    # (default call first: where the five-product form applies - causal, D = 128, S % 128 == 0, no lengths - it is what runs)
    dq1, dk1, dv1 = ops.attn_bwd2(q, k, v, o, do_masked, lse, B, S, H, D, causal, seqlens=lens_t if lens else None)
    assert relerr(dq1.view(B, S, H, D), q32.grad) < tol and relerr(dk1.view(B, S, H, D), k32.grad) < tol and relerr(dv1.view(B, S, H, D), v32.grad) < tol
    ops.attn_bwd_spill(False)  # the bit-identity chain below compares seven-product kernels (restored at the end of the test)
    dq2, dk2, dv2 = ops.attn_bwd2(q, k, v, o, do_masked, lse, B, S, H, D, causal, seqlens=lens_t if lens else None)
    assert torch.equal(dk1, dk2) and torch.equal(dv1, dv2)
    assert relerr(dv2.view(B, S, H, D), v32.grad) < tol, "dv"
    assert relerr(dk2.view(B, S, H, D), k32.grad) < tol, "dk"
    assert relerr(dq2.view(B, S, H, D), q32.grad) < tol, "dq"
    # the delta/lse scratch is reused across calls: a second call (other data) must not see stale rows
    dq3, dk3, dv3 = ops.attn_bwd2(q, k, v, o, do_masked, lse, B, S, H, D, causal, seqlens=lens_t if lens else None)
    assert torch.equal(dq3, dq2) and torch.equal(dk3, dk2) and torch.equal(dv3, dv2)
    # dK + dV from one kernel (default at D = 128: scores / dP once per tile, hand-pipelined halves, asm MFMAs) = the two single-output
    # kernels bit for bit - same products, operands and accumulation order
    try:
        ops.attn_bwd_fused_kv(False)
        dq4, dk4, dv4 = ops.attn_bwd2(q, k, v, o, do_masked, lse, B, S, H, D, causal, seqlens=lens_t if lens else None)
    finally:
        ops.attn_bwd_fused_kv(True)
    assert torch.equal(dq4, dq2) and torch.equal(dk4, dk2) and torch.equal(dv4, dv2)
    # ... and = the rounds 2-4 fused kernel (attn_bwd2_kv_k<MODE 3>, LDS-DMA tile copies): the default since round 5 is attn_bwd3_kv_k
    # (register-staged copies, three LDS stages, one barrier in the middle of a tile, one continuous fragment stream)
    try:
        ops.attn_bwd_fused_kv(1)
        dq5, dk5, dv5 = ops.attn_bwd2(q, k, v, o, do_masked, lse, B, S, H, D, causal, seqlens=lens_t if lens else None)
    finally:
        ops.attn_bwd_fused_kv(True)
        ops.attn_bwd_spill(True)
    assert torch.equal(dq5, dq2) and torch.equal(dk5, dk2) and torch.equal(dv5, dv2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,S,H", [(1, 128, 1), (2, 256, 2), (1, 640, 3), (2, 1152, 2), (1, 2048, 2)])
def test_attention_bwd_five_product_form(ops, dtype, B, S, H):
    """The causal D = 128 backward with S % 128 == 0 and no ragged lengths in its FIVE-product form (ops.attn_bwd2(..., spill=True), the
    default where it applies): the dK|dV kernel spills the unscaled dS it computes anyway, dQ = dS K is a one-product pass (attn_bwd3_dq_k).
    dK, dV bit-identical to the seven-product form; dQ against the fp32 reference with the same bound; deterministic; the fused inverse RoPE
    of dQ agrees with the seven-product form's; the dS scratch is shared between calls of different shapes."""
    D = 128
    qkv = rnd(B * S, 3 * H * D, dtype=dtype, scale=1.0, seed=S)
    q, k, v = (qkv[:, i * H * D:(i + 1) * H * D] for i in range(3))
    lens_t = torch.full((B,), S, dtype=torch.int32, device=dev())
    o, lse = ops.attn_fwd2(q, k, v, B, S, H, D, True)
    q32, k32, v32 = (t.float().reshape(B, S, H, D).clone().requires_grad_() for t in (q, k, v))
    ref = _attn_ref(q32, k32, v32, True, lens_t.long())
    do = rnd(B * S, H * D, dtype=dtype, seed=9)
    ref.backward(do.float().view(B, S, H, D))
    dq7, dk7, dv7 = ops.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, True, spill=False)
    dq5, dk5, dv5 = ops.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, True, spill=True)
    assert torch.equal(dk5, dk7) and torch.equal(dv5, dv7)
    tol = 8 * EPS16[dtype]
    e7, e5 = relerr(dq7.view(B, S, H, D), q32.grad), relerr(dq5.view(B, S, H, D), q32.grad)
    assert e5 < tol and e5 < 1.1 * e7 + 1e-5, (e5, e7)
    dq5b, dk5b, dv5b = ops.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, True, spill=True)
    assert torch.equal(dq5b, dq5) and torch.equal(dk5b, dk5) and torch.equal(dv5b, dv5)
    tab = ops.rope_table(S, D, 10000.0, dev())
    r7 = ops.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, True, rope=tab, spill=False)
    r5 = ops.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, True, rope=tab, spill=True)
    assert torch.equal(r5[1], r7[1]) and torch.equal(r5[2], r7[2])
    assert relerr(r5[0], r7[0].float()) < 2 * EPS16[dtype]
    # a ragged batch of the same shape falls back to the seven-product kernels (same entry point)
    rag = torch.tensor([S] * (B - 1) + [S - 3], dtype=torch.int32, device=dev())
    a = ops.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, True, seqlens=rag, spill=True)
    b_ = ops.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, True, seqlens=rag, spill=False)
    assert all(torch.equal(x, y) for x, y in zip(a, b_))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,S,H,D,causal", [(1, 577, 2, 64, False), (2, 300, 2, 128, True), (1, 613, 1, 128, True), (1, 70, 2, 128, True), (1, 17, 2, 64, False)])
def test_attention_never_reads_rows_past_the_batch(ops, dtype, B, S, H, D, causal):
    """The K|V / Q|dO tile copies go through buffer descriptors (csrc/attn_tiles.h stage_rows_buf): rows past the end of a batch element
    - the tail of a sequence's last, partial tile - must read as zeros whatever lies behind them in memory.  q, k, v, dO here are views of a
    buffer with 64 more rows of NaN behind the last batch element; a kernel that read one of them (NaN x 0 in an MFMA) would return NaN.
    The results must be those of the same tensors with a clean tail, bit for bit."""
    T = B * S
    g = torch.Generator(device="cuda").manual_seed(S)
    clean = torch.randn(T + 64, 3 * H * D, generator=g, device=dev()).to(dtype)
    clean[T:] = 0
    dirty = clean.clone()
    dirty[T:] = float("nan")
    do_c = torch.randn(T + 64, H * D, generator=g, device=dev()).to(dtype)
    do_c[T:] = 0
    do_d = do_c.clone()
    do_d[T:] = float("nan")

    def run(buf, dob):
        q, k, v = (buf[:T, i * H * D:(i + 1) * H * D] for i in range(3))
        o, lse = ops.attn_fwd2(q, k, v, B, S, H, D, causal)
        dq, dk, dv = ops.attn_bwd2(q, k, v, o, dob[:T], lse, B, S, H, D, causal)
        return o, dq, dk, dv

    a, b = run(clean, do_c), run(dirty, do_d)
    for x, y, name in zip(a, b, ("o", "dq", "dk", "dv")):
        assert torch.isfinite(y.float()).all(), name
        assert torch.equal(x, y), name


@pytest.mark.parametrize("dtype", DTYPES)
def test_patch_embed_pieces(ops, dtype):
    N, img, ps, d = 3, 56, 14, 128
    G = img // ps
    pix = torch.randn(N, 3, img, img, device=dev())
    kpad = 640
    cols = ops.im2col_patches(pix, ps, kpad, dtype)
    ref = torch.nn.functional.unfold(pix.to(dtype).float(), ps, stride=ps).transpose(1, 2).reshape(N * G * G, 3 * ps * ps)
    assert torch.equal(cols[:, :588].float(), ref)
    assert float(cols[:, 588:].abs().max()) == 0.0
    # token-major form used by the tower: one zero CLS slot per image, written per image tensor into a shared buffer
    S = G * G + 1
    cols2 = torch.full((N * S, kpad), 7.0, dtype=dtype, device=dev())
    ops.im2col_patches(pix[:1], ps, kpad, dtype, rows_per_img=S, row0=1, out=cols2[:S])
    ops.im2col_patches(pix[1:], ps, kpad, dtype, rows_per_img=S, row0=1, out=cols2[S:])
    c3 = cols2.view(N, S, kpad)
    assert float(c3[:, 0].abs().max()) == 0.0 and torch.equal(c3[:, 1:, :588].reshape(N * G * G, 588).float(), ref)
    patch, cls, pos = rnd(N * S, d, dtype=dtype), rnd(d, dtype=dtype, seed=1), rnd(S, d, dtype=dtype, seed=2)
    x = ops.vit_assemble(patch, cls, pos, N, G * G)
    r = torch.cat([cls.float().expand(N, 1, d), patch.float().view(N, S, d)[:, 1:]], 1) + pos.float()[None]
    assert relerr(x.view(N, S, d), r) < 2 * EPS16[dtype]
    # strided 2-D copy / accumulate (K-padding of the patch-embedding weight and of its gradient)
    w = rnd(d, 588, dtype=dtype, seed=5)
    wp = torch.zeros(d, kpad, dtype=dtype, device=dev())
    ops.copy2d(w, wp)
    assert torch.equal(wp[:, :588], w) and float(wp[:, 588:].abs().max()) == 0.0
    g = rnd(d, 588, dtype=dtype, seed=6)
    want = (g.float() + wp[:, :588].float()).to(dtype)
    ops.copy2d(wp[:, :588], g, accumulate=True)
    assert torch.equal(g, want)


def test_splice_index_and_embed(ops):
    V, P, d = 100, 4, 64
    PATCH, ST, EN = V, V + 1, V + 2
    ids = torch.tensor([[1, ST, PATCH, PATCH, PATCH, PATCH, EN, 5, 6, ST, PATCH, PATCH, PATCH, PATCH, EN, 2],
                        [1, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 2, 0],
                        [1, 3, ST, PATCH, PATCH, PATCH, PATCH, EN, 4, 2, 0, 0, 0, 0, 0, 0]], dtype=torch.int64, device=dev())
    img_off = torch.tensor([0, 2, 3, 5], dtype=torch.int32, device=dev())  # sample 2 has 2 images but 1 start (extra ignored)
    err = torch.zeros(4, dtype=torch.int32, device=dev())
    src = ops.splice_index(ids, img_off, P, PATCH, ST, EN, err)
    assert err.tolist()[:2] == [0, 0]
    exp = torch.full((3, 16), -1, dtype=torch.int32)
    exp[0, 2:6] = torch.arange(0, 4)
    exp[0, 10:14] = torch.arange(4, 8)
    exp[2, 3:7] = torch.arange(12, 16)  # image index 3 (offset of sample 2)
    assert torch.equal(src.cpu(), exp)
    # error: count mismatch
    bad = ids.clone(); bad[0, 14] = 5
    err.zero_(); ops.splice_index(bad, img_off, P, PATCH, ST, EN, err)
    assert err[0].item() == 1
    # error: misplaced end
    bad = ids.clone(); bad[2, 7], bad[2, 8] = 4, EN
    err.zero_(); ops.splice_index(bad, img_off, P, PATCH, ST, EN, err)
    assert err[1].item() == 1
    for dtype in DTYPES:
        emb, feats = rnd(V + 3, d, dtype=dtype), rnd(5 * P, d, dtype=dtype, seed=1)
        out = ops.embed_splice_fwd(ids, src, emb, feats)
        ref = emb[ids.view(-1)].clone()
        sel = src.view(-1) >= 0
        ref[sel] = feats[src.view(-1)[sel].long()]
        assert torch.equal(out, ref)
        dout = rnd(48, d, dtype=dtype, seed=2)
        dfe = torch.zeros(5 * P, d, dtype=dtype, device=dev())
        dem = torch.zeros(V + 3, d, dtype=torch.float32, device=dev())
        ops.embed_splice_bwd(ids, src, dout, dfe, dem)
        rfe = torch.zeros_like(dfe); rfe[src.view(-1)[sel].long()] = dout[sel]
        assert torch.equal(dfe, rfe)
        rem = torch.zeros_like(dem); rem.index_add_(0, ids.view(-1)[~sel], dout[~sel].float())
        assert relerr(dem, rem) < 1e-6


@pytest.mark.parametrize("V", [103, 32003])
def test_cross_entropy(ops, V):
    B, S = 2, 37
    Vpad = (V + 63) // 64 * 64
    logits = torch.randn(B * S, Vpad, device=dev()) * 3
    labels = torch.randint(0, V, (B, S), device=dev())
    labels[0, :5] = -100
    labels[1, 10:20] = -100
    row_loss, lse, out2 = ops.ce_fwd(logits, labels, V)
    lg = logits[:, :V].view(B, S, V).clone().requires_grad_()
    ref = torch.nn.functional.cross_entropy(lg[:, :-1].reshape(-1, V), labels[:, 1:].reshape(-1), ignore_index=-100)
    loss = out2[0] / out2[1]
    assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref))
    assert float(out2[1]) == float((labels[:, 1:] != -100).sum())
    ref.backward()
    for dtype in DTYPES:
        dl = ops.ce_bwd(logits, labels, lse, out2, V, Vpad, 1.0, dtype)
        assert relerr(dl[:, :V].view(B, S, V), lg.grad) < 2 * EPS16[dtype]
        assert float(dl[:, V:].abs().max()) == 0.0 if Vpad > V else True


@pytest.mark.parametrize("dtype", DTYPES)
def test_adamw(ops, dtype):
    n = 4096
    p0, g = rnd(n, dtype=dtype), rnd(n, dtype=dtype, seed=1, scale=0.01)
    ref = p0.float().clone().requires_grad_()
    opt = torch.optim.AdamW([ref], lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    m, v = torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    p = p0.clone()
    for step in range(1, 4):
        ref.grad = g.float()
        opt.step()
        ops.adamw_(p, g, m, v, 1e-2, 0.9, 0.95, 1e-8, 0.1, step)
    assert relerr(p, ref.detach()) < 6 * EPS16[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 768, 192), (1000, 520, 256), (4096, 1024, 2048), (104, 2048, 128), (4096, 11008, 512)])
def test_gemm_kstrided_operands(ops, dtype, M, N, K):
    """Operand layouts of the backward GEMMs (no transposed copies): B K-strided (dgrad), A and B K-strided
    (wgrad), A K-strided alone; identity x asymmetric checks catch a mis-ordered transpose read."""
    a, b = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, seed=1, scale=0.5)
    ref = a.float() @ b.float().t()
    at, bt = a.t().contiguous(), b.t().contiguous()  # [K, M], [K, N]
    tol = 3 * EPS16[dtype]
    nn = ops.gemm_nt(a, bt, b_t=True)
    assert nn.shape == (M, N) and relerr(nn, ref) < tol, "B K-strided"
    tn = ops.gemm_nt(at, bt, a_t=True, b_t=True)
    assert relerr(tn, ref) < tol, "A and B K-strided"
    tt = ops.gemm_nt(at, b, a_t=True)
    assert relerr(tt, ref) < tol, "A K-strided"
    for _ in range(2):
        assert torch.equal(ops.gemm_nt(at, bt, a_t=True, b_t=True), tn)
    acc = torch.ones(M, N, device=dev())
    ops.gemm_nt(at, bt, a_t=True, b_t=True, out=acc, accum=True)
    assert relerr(acc, ref + 1) < 1e-5
    if M == K:
        eye = torch.eye(K, dtype=dtype, device=dev())
        asym = (torch.arange(N * K, device=dev()).reshape(K, N) % 251).to(dtype)  # B[k, n]
        assert torch.equal(ops.gemm_nt(eye, asym, b_t=True).float(), asym.float())
        assert torch.equal(ops.gemm_nt(eye, asym, a_t=True, b_t=True).float(), asym.float())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (304, 520, 256), (1000, 264, 384), (512, 768, 2048), (2064, 1288, 128)])
def test_gemm_w4_all_layouts_and_epilogues(ops, dtype, M, N, K):
    """csrc/gemm_w4.hip (4 waves x 128x128 outputs, mh_gemm_force_kernel(4)) in its NT / NN / TN / TT operand forms, edge tiles in both
    dimensions, every staged epilogue incl. accumulate - against fp32 torch and against the 8-wave kernel (force 256); identity x
    asymmetric operands catch a transposed fragment or C write (cdna guide rule 16)."""
    a, b = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, seed=1, scale=0.5)
    bias, resid = rnd(N, dtype=dtype, seed=2), rnd(M, N, dtype=dtype, seed=3)
    ref = a.float() @ b.float().t()
    at, bt = a.t().contiguous(), b.t().contiguous()
    tol = 3 * EPS16[dtype]
    try:
        res = {}
        for which in (4, 256):
            ops.gemm_force_kernel(which)
            r = res[which] = {}
            r["nt"] = ops.gemm_nt(a, b)
            r["nn"] = ops.gemm_nt(a, bt, b_t=True)
            r["tn"] = ops.gemm_nt(at, bt, a_t=True, b_t=True)
            r["tt"] = ops.gemm_nt(at, b, a_t=True)
            r["bias"] = ops.gemm_nt(a, b, bias=bias)
            r["gelu"] = ops.gemm_nt(a, b, bias=bias, act="quick_gelu")
            r["resid"] = ops.gemm_nt(a, bt, b_t=True, resid=resid)
            r["bias_resid"] = ops.gemm_nt(a, b, bias=bias, resid=resid)
            acc = rnd(M, N, dtype=dtype, seed=4)
            ops.gemm_nt(at, bt, a_t=True, b_t=True, out=acc, accum=True)
            r["accum"] = acc
        z = ref + bias.float()
        want = dict(nt=ref, nn=ref, tn=ref, tt=ref, bias=z, gelu=z * torch.sigmoid(1.702 * z), resid=ref + resid.float(),
                    bias_resid=z + resid.float(), accum=ref + rnd(M, N, dtype=dtype, seed=4).float())
        for k, w in want.items():
            assert relerr(res[4][k], w) < (4 * EPS16[dtype] if k != "nt" else tol), k
            assert relerr(res[4][k], res[256][k].float()) < 2 * EPS16[dtype], (k, "vs the 8-wave kernel")
        ops.gemm_force_kernel(4)
        assert torch.equal(ops.gemm_nt(at, bt, a_t=True, b_t=True), res[4]["tn"])  # deterministic
        if M == N == 256:
            eye = torch.eye(256, dtype=dtype, device=dev())
            asym = (torch.arange(256 * 256, device=dev()).reshape(256, 256) % 251).to(dtype)
            assert torch.equal(ops.gemm_nt(eye, asym).float(), asym.float().t())                    # out[m, n] = asym[n, m]
            assert torch.equal(ops.gemm_nt(eye, asym, b_t=True).float(), asym.float())              # B[k, n]
            assert torch.equal(ops.gemm_nt(eye, asym, a_t=True, b_t=True).float(), asym.float())
            assert torch.equal(ops.gemm_nt(asym, eye, a_t=True).float(), asym.float().t())          # A[k, m] -> out[m, n = k]
    finally:
        ops.gemm_force_kernel(0)


def test_gemm_w4_at_the_weight_gradient_geometry(ops):
    """The shapes the auto selection sends to gemm_w4: TN weight gradients over 32 768 tokens (q|k|v, gate|up with 86 tile columns,
    down) - sampled tiles vs fp32, and the same launch through wgrad_tn."""
    T = 32768
    for No, Ki in ((12288, 4096), (4096, 11008)):
        dy, x = rnd(T, No, dtype=torch.bfloat16, scale=0.05), rnd(T, Ki, dtype=torch.bfloat16, seed=1)
        out = torch.empty(No, Ki, dtype=torch.bfloat16, device=dev())
        ops.wgrad_tn(dy, x, out, accum=False)
        for (r0, c0) in ((0, 0), (No - 256, Ki - 256), (1024 + 128, 2048 + 64)):
            ref = dy[:, r0:r0 + 256].float().t() @ x[:, c0:c0 + 256].float()
            assert relerr(out[r0:r0 + 256, c0:c0 + 256], ref) < 3 * EPS16[torch.bfloat16], (No, Ki, r0, c0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,No,Ki", [(27696, 1024, 1024), (27696, 3072, 1024), (1731, 1024, 256), (130, 512, 264), (64, 256, 256),
                                     (4100, 4096, 4096), (8192, 1024, 4096), (37, 256, 8)])
def test_wgrad_tn_any_tokens_splitk(ops, dtype, T, No, Ki):
    """Weight gradient dW[No, Ki] (+)= dY[T, No]^T X[T, Ki] with the token count as the contraction: any T (zero-row
    tail of the last K-tile), split-K for outputs too small to fill the chip (CLIP tower shapes), fresh and
    accumulating, bit-identical run to run (fixed reduction order)."""
    dy, x = rnd(T, No, dtype=dtype, scale=0.5), rnd(T, Ki, dtype=dtype, seed=1, scale=0.5)
    ref = dy.float().t() @ x.float()
    out = torch.empty(No, Ki, dtype=dtype, device=dev())
    ops.wgrad_tn(dy, x, out, accum=False)
    assert relerr(out, ref) < 3 * EPS16[dtype]
    first = out.clone()
    for _ in range(2):
        ops.wgrad_tn(dy, x, out, accum=False)
        assert torch.equal(out, first)
    old = rnd(No, Ki, dtype=dtype, seed=5)
    acc = old.clone()
    ops.wgrad_tn(dy, x, acc, accum=True)
    assert relerr(acc, ref + old.float()) < 4 * EPS16[dtype]
    acc32 = torch.ones(No, Ki, device=dev())
    ops.wgrad_tn(dy, x, acc32, accum=True)
    assert relerr(acc32, ref + 1) < 2e-5
    # a strided view of a wider buffer (fused q|k|v gradient spans are written as views)
    wide = torch.zeros(No, Ki + 64, dtype=dtype, device=dev())
    ops.wgrad_tn(dy, x, wide[:, 8:8 + Ki], accum=False)
    assert torch.equal(wide[:, 8:8 + Ki], first) and float(wide[:, :8].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 256, 128), (512, 768, 192), (1000, 515, 256), (613, 4096, 1024),
                                   (2048, 2048, 4096), (300, 103, 64), (4096, 11008, 512)])
@pytest.mark.parametrize("which", [256])
def test_gemm_256_tile_kernel(ops, dtype, M, N, K, which):
    """The pipelined 256x256 kernel (forced), incl.
    M/N edges, short K (prologue/tail clamps) and epilogues (direct and LDS-staged); repeated to catch pipeline
    races (results must be bit-identical run to run)."""
    a, b = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, seed=1, scale=0.5)
    bias, resid = rnd(N, dtype=dtype, seed=2), rnd(M, N, dtype=dtype, seed=3)
    ref = a.float() @ b.float().t()
    try:
        ops.gemm_force_kernel(which)
        out = ops.gemm_nt(a, b)
        assert relerr(out, ref) < 3 * EPS16[dtype]
        for _ in range(3):
            assert torch.equal(ops.gemm_nt(a, b), out)
        o32 = ops.gemm_nt(a, b, out_f32=True)
        assert relerr(o32, ref) < 1e-5
        if N % 4 == 0:
            z = ops.gemm_nt(a, b, bias=bias, resid=resid)
            assert relerr(z, ref + bias.float() + resid.float()) < 4 * EPS16[dtype]
            z = ops.gemm_nt(a, b, resid=resid)
            assert relerr(z, ref + resid.float()) < 4 * EPS16[dtype]
            z = ops.gemm_nt(a, b, bias=bias, act="quick_gelu")
            y = ref + bias.float()
            assert relerr(z, y * torch.sigmoid(1.702 * y)) < 4 * EPS16[dtype]
        ops.gemm_force_kernel(128)
        assert torch.equal(ops.gemm_nt(a, b, out_f32=True), o32) or relerr(ops.gemm_nt(a, b, out_f32=True), o32) < 1e-5
    finally:
        ops.gemm_force_kernel(0)


# ---- KV-cache decode kernels ---------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(1, 4096, 4096), (8, 12288, 4096), (3, 515, 264), (5, 32003, 256), (11, 1024, 11008), (16, 4096, 4096), (4, 22016, 4096),
                                   (13, 515, 11008), (20, 1000, 2080)])
def test_gemv_small_m(ops, dtype, M, N, K):
    """Both GEMV forms (one wave per weight row; MFMA with the weights streamed into the B-operand registers, 3..16 rows) at every M."""
    x, w = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, seed=1, scale=0.5)
    resid = rnd(M, N, dtype=dtype, seed=2)
    ref = x.float() @ w.float().t()
    try:
        for mn in ((3, 17) if M <= 8 else (3,)):
            ops.gemv_mfma_min_rows(mn)
            assert relerr(ops.gemv(x, w), ref) < 3 * EPS16[dtype]
            assert relerr(ops.gemv(x, w, resid=resid), ref + resid.float()) < 3 * EPS16[dtype]
            assert relerr(ops.gemv(x, w, out_f32=True), ref) < 1e-5
            assert torch.equal(ops.gemv(x, w), ops.gemv(x, w))
            # against the MFMA GEMM on the same operands (the prefill path): same values up to the summation order
            if K % 64 == 0:
                assert relerr(ops.gemv(x, w, out_f32=True), ops.gemm_nt(x, w, out_f32=True)) < 1e-5
    finally:
        ops.gemv_mfma_min_rows(0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(12, 264, 264), (16, 520, 72), (23, 128, 40)])
def test_gemv_9_to_16_rows_without_the_mfma_form(ops, dtype, M, N, K):
    """9-16 activation rows exist only in the MFMA form (K % 32 == 0, row threshold at its default); the wrappers
    chunk by 8 rows when that form is out of reach (K a multiple of 8 only, or mh_gemv_mfma_min_rows(17)) instead of raising."""
    x, w = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, seed=1, scale=0.5)
    ref = x.float() @ w.float().t()
    assert relerr(ops.gemv(x, w), ref) < 3 * EPS16[dtype]
    qw = ops.quant_fp8_b128(w) if K % 128 == 0 else None
    try:
        ops.gemv_mfma_min_rows(17)
        x2, w2 = rnd(M, 256, dtype=dtype), rnd(N, 256, dtype=dtype, seed=1, scale=0.5)
        assert relerr(ops.gemv(x2, w2), x2.float() @ w2.float().t()) < 3 * EPS16[dtype]
        q2 = ops.quant_fp8_b128(w2)
        assert relerr(ops.gemv_fp8w(x2, q2), x2.float() @ w2.float().t()) < 0.08
    finally:
        ops.gemv_mfma_min_rows(0)
    del qw


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,ff,K", [(1, 11008, 4096), (2, 520, 264), (3, 1000, 4096), (5, 11008, 4096), (8, 136, 2048), (11, 256, 512)])
def test_gemv_swiglu_fused(ops, dtype, M, ff, K):
    """Decode-step gate|up projection + SwiGLU in one launch = the two launches (gate / up rounded to 16 bits before the activation;
    the activation itself may round differently in the last bit), and HF LlamaMLP's act_fn(gate_proj(x)) * up_proj(x) in fp32."""
    x, wgu = rnd(M, K, dtype=dtype), rnd(2 * ff, K, dtype=dtype, seed=1, scale=0.1)
    try:
        if M <= 8:
            ops.gemv_mfma_min_rows(17)  # both sides on the row-per-wave GEMV form (the MFMA form: test_gemv_swiglu_fused_mfma_form)
        one = ops.gemv_swiglu(x, wgu)
        if M <= 8:
            two = ops.swiglu_fwd(ops.gemv(x, wgu))
            assert float((one != two).float().mean()) < 1e-3 and relerr(one, two) < EPS16[dtype]
    finally:
        ops.gemv_mfma_min_rows(0)
    gu = (x.float() @ wgu.float().t()).to(dtype).float()  # the projection as the reference stores it
    ref = torch.nn.functional.silu(gu[:, :ff]) * gu[:, ff:]
    assert relerr(one, ref) < 4 * EPS16[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(1, 12288, 4096), (2, 528, 264), (4, 22016, 4096), (5, 1008, 1024), (8, 4096, 4096), (12, 256, 512)])
def test_gemv_with_fused_rmsnorm(ops, dtype, M, N, K):
    """input_layernorm / post_attention_layernorm folded into the projection launch of the decode step = rmsnorm_fwd + gemv bit for bit
    (+ swiglu within the activation's last bit), and the fp32 torch expression within 16-bit rounding."""
    x, g, w = rnd(M, K, dtype=dtype), rnd(K, dtype=dtype, seed=3), rnd(N, K, dtype=dtype, seed=1, scale=0.1)
    h = ops.rmsnorm_fwd(x, g, 1e-6)
    keep = ops.FUSED_NORM_MAX_ROWS
    try:
        ops.FUSED_NORM_MAX_ROWS = 8  # exercise the fused kernel at every row count it supports
        if M <= 8:
            ops.gemv_mfma_min_rows(17)
            assert torch.equal(ops.gemv_norm(x, g, 1e-6, w), ops.gemv(h, w))
            one, two = ops.gemv_norm(x, g, 1e-6, w, swiglu=True), ops.swiglu_fwd(ops.gemv(h, w))
            assert float((one != two).float().mean()) < 1e-3 and relerr(one, two) < EPS16[dtype]
        xf = x.float()
        hn = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * g.float()).to(dtype).float()
        assert relerr(ops.gemv_norm(x, g, 1e-6, w), hn @ w.float().t()) < 4 * EPS16[dtype]
    finally:
        ops.gemv_mfma_min_rows(0)
        ops.FUSED_NORM_MAX_ROWS = keep


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(1, 12288, 4096), (2, 528, 512), (4, 22016, 4096), (8, 1024, 2048), (11, 256, 512)])
def test_gemv_fp8w_with_fused_rmsnorm_and_swiglu(ops, dtype, M, N, K):
    """fp8-weight decode projections with the RMSNorm (1-2 rows) and SwiGLU (up to 8 rows) folded in = the separate launches."""
    x, g, w = rnd(M, K, dtype=dtype), rnd(K, dtype=dtype, seed=3), rnd(N, K, dtype=dtype, seed=1, scale=0.1)
    qw = ops.quant_fp8_b128(w)
    h = ops.rmsnorm_fwd(x, g, 1e-6)
    keep = ops.FUSED_NORM_MAX_ROWS
    try:
        ops.FUSED_NORM_MAX_ROWS = 8
        if M <= 8:
            ops.gemv_mfma_min_rows(17)
        ref = ops.gemv_fp8w(h, qw)
        got = ops.gemv_fp8w_norm(x, g, 1e-6, qw)
        assert torch.equal(got, ref) if M <= 8 else relerr(got, ref) < 2 * EPS16[dtype]
        one, two = ops.gemv_fp8w_norm(x, g, 1e-6, qw, swiglu=True), ops.swiglu_fwd(ref)
        assert float((one != two).float().mean()) < 1e-3 and relerr(one, two) < 2 * EPS16[dtype]
    finally:
        ops.gemv_mfma_min_rows(0)
        ops.FUSED_NORM_MAX_ROWS = keep


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("B,H,D,K,Smax,pos", [(1, 32, 128, 4096, 64, [17]), (2, 4, 64, 256, 40, [0, 39]), (5, 2, 128, 512, 16, [3, 0, 15, 7, 7]),
                                              (8, 4, 128, 1024, 24, list(range(8))), (11, 2, 64, 256, 12, list(range(11)))])
def test_qkv_projection_with_fused_norm_rope_append(ops, dtype, fp8, B, H, D, K, Smax, pos):
    """input_layernorm + q|k|v projection + RoPE + K/V append of one decode step in one launch = the four separate launches, bit for bit
    (qkv buffer and the touched cache rows; untouched cache rows stay as they were)."""
    d = H * D
    x, g, w = rnd(B, K, dtype=dtype), rnd(K, dtype=dtype, seed=3), rnd(3 * d, K, dtype=dtype, seed=1, scale=0.1)
    wq = ops.quant_fp8_b128(w) if fp8 else w
    tab = ops.rope_table(Smax, D, 10000.0, dev())
    p32 = torch.tensor(pos, dtype=torch.int32, device=dev())
    kc0, vc0 = rnd(B, Smax, d, dtype=dtype, seed=5), rnd(B, Smax, d, dtype=dtype, seed=6)
    keep = ops.FUSED_NORM_MAX_ROWS
    try:
        if B <= 8:
            ops.gemv_mfma_min_rows(17)
        h = ops.rmsnorm_fwd(x, g, 1e-6)
        ref = ops.gemv_fp8w(h, wq) if fp8 else ops.gemv(h, w)
        kc1, vc1 = kc0.clone(), vc0.clone()
        ops.decode_rope_append(ref, tab, p32, kc1, vc1, H, D)
        for fuse_rows in (8, 0):  # norm inside the launch / separate
            ops.FUSED_NORM_MAX_ROWS = fuse_rows
            kc2, vc2 = kc0.clone(), vc0.clone()
            got = ops.gemv_qkv_rope(x, g, 1e-6, wq, tab, p32, kc2, vc2, H, D)
            assert torch.equal(got, ref) and torch.equal(kc2, kc1) and torch.equal(vc2, vc1)
    finally:
        ops.gemv_mfma_min_rows(0)
        ops.FUSED_NORM_MAX_ROWS = keep


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("M,ff,K", [(3, 11008, 4096), (4, 136, 256), (8, 1000, 4096), (11, 11008, 4096), (16, 520, 1024)])
def test_gemv_swiglu_fused_mfma_form(ops, dtype, fp8, M, ff, K):
    """3-16 rows: the gate|up projection runs as the MFMA GEMV whose blocks own 16 gate rows AND their up rows and apply SwiGLU in the
    epilogue = MFMA GEMV (8-wave form: same summation order) + swiglu_fwd up to the activation's last bit."""
    x, wgu = rnd(M, K, dtype=dtype), rnd(2 * ff, K, dtype=dtype, seed=1, scale=0.1)
    qw = ops.quant_fp8_b128(wgu) if fp8 else None
    try:
        ops.gemv_mfma_pair_min_rows(3, 3)  # (defaults 6 / 4: below that the one-wave-per-row-pair form is used)
        one = ops.gemv_fp8w_norm(x, rnd(K, dtype=dtype, seed=3), 1e-6, qw, swiglu=True) if fp8 else ops.gemv_swiglu(x, wgu)
        ops.gemv_mfma_wide(False)
        if fp8:
            two = ops.swiglu_fwd(ops.gemv_fp8w(ops.rmsnorm_fwd(x, rnd(K, dtype=dtype, seed=3), 1e-6), qw))
        else:
            two = ops.swiglu_fwd(ops.gemv(x, wgu))
    finally:
        ops.gemv_mfma_wide(True)
        ops.gemv_mfma_pair_min_rows(0, 0)
    assert one.shape == (M, ff)
    assert int((one != two).sum()) <= max(2, one.numel() // 1000) and relerr(one, two) < EPS16[dtype]  # (last bit of the activation)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("B,H,D,K,Smax,pos", [(3, 32, 128, 4096, 64, [17, 0, 63]), (5, 2, 64, 256, 16, [3, 0, 15, 7, 7]),
                                              (8, 4, 128, 1024, 24, list(range(8))), (16, 2, 128, 512, 20, list(range(16)))])
def test_qkv_projection_rope_append_mfma_form(ops, dtype, fp8, B, H, D, K, Smax, pos):
    """3-16 rows: q|k|v projection + RoPE + K/V append as the MFMA GEMV whose blocks own 16 channels c of a head and their partners
    c + D/2 = rmsnorm + MFMA GEMV (8-wave form) + decode_rope_append, bit for bit (qkv buffer and cache)."""
    d = H * D
    x, g, w = rnd(B, K, dtype=dtype), rnd(K, dtype=dtype, seed=3), rnd(3 * d, K, dtype=dtype, seed=1, scale=0.1)
    wq = ops.quant_fp8_b128(w) if fp8 else w
    tab = ops.rope_table(Smax, D, 10000.0, dev())
    p32 = torch.tensor(pos, dtype=torch.int32, device=dev())
    kc0, vc0 = rnd(B, Smax, d, dtype=dtype, seed=5), rnd(B, Smax, d, dtype=dtype, seed=6)
    h = ops.rmsnorm_fwd(x, g, 1e-6)
    try:
        ops.gemv_mfma_wide(False)
        ref = ops.gemv_fp8w(h, wq) if fp8 else ops.gemv(h, w)
    finally:
        ops.gemv_mfma_wide(True)
    kc1, vc1 = kc0.clone(), vc0.clone()
    ops.decode_rope_append(ref, tab, p32, kc1, vc1, H, D)
    kc2, vc2, kc3, vc3 = kc0.clone(), vc0.clone(), kc0.clone(), vc0.clone()
    try:
        ops.gemv_mfma_pair_min_rows(3, 3)  # (defaults 6 / 4: below that the one-wave-per-row-pair form is used)
        ops.gemv_mfma_wide(False)
        got = ops.gemv_qkv_rope(x, g, 1e-6, wq, tab, p32, kc2, vc2, H, D)
        ops.gemv_mfma_wide(True)
        # the default form splits K over 16 waves (another summation order): the same values within 16-bit rounding
        got16 = ops.gemv_qkv_rope(x, g, 1e-6, wq, tab, p32, kc3, vc3, H, D)
    finally:
        ops.gemv_mfma_wide(True)
        ops.gemv_mfma_pair_min_rows(0, 0)
    assert torch.equal(got, ref) and torch.equal(kc2, kc1) and torch.equal(vc2, vc1)
    assert relerr(got16, ref) < EPS16[dtype] and relerr(kc3, kc1) < EPS16[dtype] and relerr(vc3, vc1) < EPS16[dtype]
    untouched = torch.ones(B, Smax, dtype=torch.bool, device=dev())
    untouched[torch.arange(B, device=dev()), p32.long()] = False
    assert torch.equal(kc3[untouched], kc0[untouched]) and torch.equal(vc3[untouched], vc0[untouched])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,d", [(1, 4096), (3, 1024), (16, 4096), (5, 8192), (64, 264)])
def test_rmsnorm_few_rows(ops, dtype, rows, d):
    """The block-per-row RMSNorm the decode step uses (<= 64 rows) against torch fp32 (HF LlamaRMSNorm)."""
    x, w = rnd(rows, d, dtype=dtype), rnd(d, dtype=dtype, seed=1)
    y = ops.rmsnorm_fwd(x, w, 1e-6)
    xf = x.float()
    ref = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * w.float()
    assert relerr(y, ref) < 2 * EPS16[dtype]
    assert torch.equal(y, ops.rmsnorm_fwd(x, w, 1e-6))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,D,Smax,lens", [(2, 2, 128, 40, [17, 40]), (3, 4, 64, 700, [1, 333, 700]), (1, 32, 128, 4200, [4100])])
def test_decode_rope_append_and_attention(ops, dtype, B, H, D, Smax, lens):
    """One decode step against a torch fp32 restatement of HF LlamaAttention with a KV cache: rotate-half RoPE at each
    sequence's own position, append, softmax(q K^T / sqrt(D)) V over the valid keys."""
    d = H * D
    kc, vc = rnd(B, Smax, d, dtype=dtype, seed=3), rnd(B, Smax, d, dtype=dtype, seed=4)
    qkv = rnd(B, 3 * d, dtype=dtype, seed=5)
    pos = torch.tensor([l - 1 for l in lens], dtype=torch.int32, device=dev())   # the new token's position
    tab = ops.rope_table(Smax, D, 10000.0, dev())
    kc0, vc0, qkv0 = kc.clone(), vc.clone(), qkv.clone()
    ops.decode_rope_append(qkv, tab, pos, kc, vc, H, D)
    # reference rotation
    def rot(x, p):  # x [H, D] fp32
        cos, sin = tab[p, :, 0], tab[p, :, 1]
        lo, hi = x[:, :D // 2], x[:, D // 2:]
        return torch.cat([lo * cos - hi * sin, hi * cos + lo * sin], dim=1)
    for b in range(B):
        p_ = int(pos[b])
        qr = rot(qkv0[b, :d].float().view(H, D), p_)
        kr = rot(qkv0[b, d:2 * d].float().view(H, D), p_)
        assert relerr(qkv[b, :d].float().view(H, D), qr) < 2 * EPS16[dtype]
        assert relerr(kc[b, p_].float().view(H, D), kr) < 2 * EPS16[dtype]
        assert torch.equal(vc[b, p_], qkv0[b, 2 * d:])
        keep = torch.ones(Smax, dtype=torch.bool, device=dev()); keep[p_] = False
        assert torch.equal(kc[b][keep], kc0[b][keep]) and torch.equal(vc[b][keep], vc0[b][keep])
    lens_t = torch.tensor(lens, dtype=torch.int32, device=dev())
    o = ops.attn_decode(qkv[:, :d], kc, vc, lens_t, H, D)
    o1 = ops.attn_decode(qkv[:, :d], kc, vc, lens_t, H, D, split_kv=False)  # one block per (b, h)
    assert relerr(o1, o.float()) < 2 * EPS16[dtype]
    try:  # the split-KV merge by the last block of a (b, h) (A/B arm) against the merge as a second launch: same arithmetic, same bits
        ops.attn_decode_fused_merge(True)
        for _ in range(3):  # (the ticket counters are left at zero by every launch)
            assert torch.equal(ops.attn_decode(qkv[:, :d], kc, vc, lens_t, H, D), o)
    finally:
        ops.attn_decode_fused_merge(False)
    for b in range(B):
        L_ = lens[b]
        qh = qkv[b, :d].float().view(H, 1, D)
        kh = kc[b, :L_].float().view(L_, H, D).permute(1, 0, 2)
        vh = vc[b, :L_].float().view(L_, H, D).permute(1, 0, 2)
        ref = torch.softmax(qh @ kh.transpose(1, 2) / D ** 0.5, dim=-1) @ vh
        assert relerr(o[b].float().view(H, 1, D), ref) < 3 * EPS16[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,S,H,D,K", [(2, 96, 2, 128, 256), (1, 613, 32, 128, 4096), (3, 50, 4, 64, 256), (2, 256, 1, 128, 128)])
def test_gemm_with_fused_rope_is_bit_identical_to_gemm_then_rope(ops, dtype, B, S, H, D, K):
    """The fused q|k|v projection + RoPE (staged GEMM epilogue) must reproduce mh_gemm_nt followed by mh_rope_qk bit for
    bit: q and k heads rotated at position (row % S), v untouched; row counts that are not a multiple of the tile."""
    T, d = B * S, H * D
    x, w = rnd(T, K, dtype=dtype), rnd(3 * d, K, dtype=dtype, seed=1, scale=0.5)
    tab = ops.rope_table(S, D, 10000.0, dev())
    try:
        ops.gemm_force_kernel(256)  # the fused form lives in the 256-tile kernel: same summation order for the reference
        ref = ops.gemm_nt(x, w)
    finally:
        ops.gemm_force_kernel(0)
    ops.rope_qk_(ref, tab, S, H, D)
    try:
        ops.gemm_force_kernel(256)  # (the 4-wave kernel rotates the fp32 accumulators instead: tests/test_gemm_w4_gpu.py)
        got = ops.gemm_nt_rope(x, w, tab, S, H, D)
        assert torch.equal(got, ref)
        for _ in range(2):
            assert torch.equal(ops.gemm_nt_rope(x, w, tab, S, H, D), got)
    finally:
        ops.gemm_force_kernel(0)
    assert relerr(ops.gemm_nt_rope(x, w, tab, S, H, D), ref.float()) < 3 * EPS16[dtype]  # whichever kernel the dispatcher picks


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,S,H,D,causal", [(2, 200, 2, 128, True), (1, 577, 4, 64, False), (2, 256, 3, 128, True)])
def test_attention_bwd_with_fused_inverse_rope(ops, dtype, B, S, H, D, causal):
    """dq, dk rotated back inside the backward kernels' epilogues (fp32, before the single rounding) against the
    separate path: attention backward, then the stand-alone inverse RoPE on the stored 16-bit gradients."""
    d = H * D
    qkv = rnd(B * S, 3 * d, dtype=dtype, scale=0.5)
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    o, lse = ops.attn_fwd2(q, k, v, B, S, H, D, causal)
    do = rnd(B * S, d, dtype=dtype, seed=7, scale=0.5)
    tab = ops.rope_table(S, D, 10000.0, dev())
    dq, dk, dv = ops.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, causal)
    ref = torch.cat([dq, dk, dv], dim=1).contiguous()
    ops.rope_qk_(ref, tab, S, H, D, inverse=True)
    dq2, dk2, dv2 = ops.attn_bwd2(q, k, v, o, do, lse, B, S, H, D, causal, rope=tab)
    assert torch.equal(dv2, dv)
    assert relerr(dq2, ref[:, :d].float()) < 2 * EPS16[dtype]
    assert relerr(dk2, ref[:, d:2 * d].float()) < 2 * EPS16[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,ff,K", [(300, 256, 128), (1000, 1408, 256), (613, 11008, 4096), (4096, 512, 256)])
def test_gemm_with_fused_swiglu_is_bit_identical_to_unfused(ops, dtype, M, ff, K):
    """SwiGLU in the GEMM epilogues (forward: gate|up projection also emits act; backward: the down-projection dgrad emits
    dgu without storing dact) against GEMM + stand-alone SwiGLU kernels on the stored 16-bit tensors: bit for bit."""
    x, wgu = rnd(M, K, dtype=dtype), rnd(2 * ff, K, dtype=dtype, seed=1, scale=0.5)
    try:
        ops.gemm_force_kernel(256)
        gu_ref = ops.gemm_nt(x, wgu)
    finally:
        ops.gemm_force_kernel(0)
    act_ref = ops.swiglu_fwd(gu_ref)
    # (the 8-wave kernel's staged form: it gates the ROUNDED 16-bit tile, hence bit-identity with the unfused kernels; the 4-wave kernel's
    # form gates the fp32 accumulators - one rounding less - and is held to fp32 in tests/test_gemm_w4_gpu.py)
    try:
        ops.gemm_force_kernel(256)
        gu, act = ops.gemm_swiglu_fwd(x, wgu)
        assert torch.equal(gu, gu_ref) and torch.equal(act, act_ref)
        # backward: dy [M, d], wd [d, ff]
        d = K
        dy, wd = rnd(M, d, dtype=dtype, seed=2, scale=0.5), rnd(d, ff, dtype=dtype, seed=3, scale=0.5)
        dact = ops.gemm_nt(dy, wd, b_t=True)
        dgu_ref = ops.swiglu_bwd(gu_ref, dact)
        dgu = ops.gemm_swiglu_bwd(dy, wd, gu_ref)
        assert torch.equal(dgu, dgu_ref)
    finally:
        ops.gemm_force_kernel(0)
    # whichever kernel the dispatcher picks for this shape: within one rounding of the unfused result
    gu2, act2 = ops.gemm_swiglu_fwd(x, wgu)
    assert relerr(gu2, gu_ref.float()) < 2 * EPS16[dtype] and relerr(act2, act_ref.float()) < 3 * EPS16[dtype]
    assert relerr(ops.gemm_swiglu_bwd(dy, wd, gu_ref), dgu_ref.float()) < 3 * EPS16[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", [8, 4099, 1 << 20, (1 << 22) + 5])
def test_sumsq_deterministic(ops, dtype, n):
    g = rnd(n + 8, dtype=dtype, scale=0.3)[:n]  # (any length; 16-byte aligned base)
    part, out = torch.zeros(2048, device=dev()), torch.zeros(1, device=dev())
    ops.sumsq_det(g, part, out)
    ref = float((g.double() ** 2).sum())
    assert abs(float(out) - ref) / ref < 1e-5
    first = out.clone()
    for _ in range(3):
        ops.sumsq_det(g, part, out)
        assert torch.equal(out, first)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("causal", [True, False])
def test_attention_fwd_lazy_rescale_with_growing_row_maxima(ops, dtype, causal):
    """Scores whose row maximum keeps growing along the key axis by far more than the lazy-rescale threshold (2^8), and
    rows whose maximum sits in the first tile: the running reference must move exactly when needed."""
    B, S, H, D = 1, 512, 2, 128
    d = H * D
    g = torch.Generator(device="cpu").manual_seed(11)
    q = torch.randn(B * S, d, generator=g)
    k = torch.randn(B * S, d, generator=g)
    ramp = torch.linspace(0.2, 6.0, S)[:, None]           # later keys have much larger norms -> growing maxima
    k = k * ramp
    k[5] *= 8.0                                           # an early dominant key (maximum in the first tile for rows >= 5)
    v = torch.randn(B * S, d, generator=g)
    qkv = torch.cat([q, k, v], dim=1).to(dtype).to(dev()).contiguous()
    q_, k_, v_ = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    o, lse = ops.attn_fwd2(q_, k_, v_, B, S, H, D, causal)
    qh = q_.float().view(S, H, D).permute(1, 0, 2)
    kh = k_.float().view(S, H, D).permute(1, 0, 2)
    vh = v_.float().view(S, H, D).permute(1, 0, 2)
    sc = qh @ kh.transpose(1, 2) / D ** 0.5
    if causal:
        sc = sc.masked_fill(torch.ones(S, S, device=dev()).triu(1).bool(), float("-inf"))
    ref = (torch.softmax(sc, dim=-1) @ vh).permute(1, 0, 2).reshape(S, d)
    assert relerr(o, ref) < 4 * EPS16[dtype]
    lse_ref = torch.logsumexp(sc, dim=-1)                 # [H, S]
    assert float((lse[0, :, :S] - lse_ref).abs().max()) < 2e-2 * max(1.0, float(lse_ref.abs().max()) * 0.05)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(1, 4096, 4096), (8, 1000, 1024), (3, 515, 272), (2, 256, 11008), (4, 4096, 11008), (16, 12288, 4096), (13, 515, 2112)])
def test_fp8_block_quant_and_gemv(ops, dtype, M, N, K):
    """fp8 weight path of the decode step: the quantiser must produce OCP e4m3 bytes (checked by reinterpreting them as
    torch.float8_e4m3fn) with scale = max|block| / 448, and the GEMV must equal x @ dequant(q, s)^T to fp32 accuracy; the
    quantisation error itself is bounded by the format (3 mantissa bits: <= 2^-4 relative per element)."""
    w = rnd(N, K, dtype=dtype, scale=0.05)
    w[0, :128] = 0                                        # an all-zero block
    x = rnd(M, K, dtype=dtype, seed=1)
    q, s = ops.quant_fp8_b128(w)
    nb = (K + 127) // 128
    assert q.shape == (N, K) and s.shape == (N, nb)
    wpad = torch.nn.functional.pad(w.float(), (0, nb * 128 - K)).view(N, nb, 128)
    s_ref = wpad.abs().amax(dim=2) / 448.0
    s_ref = torch.where(s_ref > 0, s_ref, torch.ones_like(s_ref))
    assert torch.allclose(s, s_ref, rtol=1e-6, atol=0)
    deq = q.view(torch.float8_e4m3fn).float()             # the bytes ARE OCP e4m3
    deq = (torch.nn.functional.pad(deq, (0, nb * 128 - K)).view(N, nb, 128) * s[:, :, None]).view(N, nb * 128)[:, :K]
    err = (deq - w.float()).abs()
    assert float((err / (wpad.abs().amax(dim=2)[:, :, None].expand(N, nb, 128).reshape(N, nb * 128)[:, :K] + 1e-30)).max()) <= 2.0 ** -4 + 1e-6
    # the torch rounding of the same scaled values gives the same bytes (round-to-nearest-even, no saturation surprises)
    q_ref = (wpad * (1.0 / s_ref)[:, :, None]).view(N, nb * 128)[:, :K].to(torch.float8_e4m3fn)
    assert torch.equal(q.view(torch.float8_e4m3fn).float(), q_ref.float())
    ref = x.float() @ deq.t()
    resid = rnd(M, N, dtype=dtype, seed=3)
    try:
        for mn in ((3, 17) if M <= 8 else (3,)):  # the MFMA form and the one-wave-per-row form
            ops.gemv_mfma_min_rows(mn)
            got = ops.gemv_fp8w(x, (q, s), out_f32=True)
            assert relerr(got, ref) < 2e-5
            assert relerr(ops.gemv_fp8w(x, (q, s), resid=resid), ref + resid.float()) < 3 * EPS16[dtype]
    finally:
        ops.gemv_mfma_min_rows(0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 768, 384), (1000, 520, 256), (613, 4096, 1024), (4096, 4096, 4096), (300, 104, 128)])
def test_gemm_fp8_scaled_mfma(ops, dtype, M, N, K):
    """fp8 x fp8 GEMM on the gfx950 scaled-fp8 MFMA (per-row scales): exact against the fp32 product of the DEQUANTISED
    operands (only the summation order and the output rounding differ), within the format's error of the unquantised
    product, bit-identical run to run, with bias / residual epilogues and ragged M, N."""
    a, b = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, seed=1, scale=0.5)
    qa, qb = ops.quant_fp8_rows(a), ops.quant_fp8_rows(b)
    da = qa[0].view(torch.float8_e4m3fn).float() * qa[1][:, None]
    db = qb[0].view(torch.float8_e4m3fn).float() * qb[1][:, None]
    assert torch.allclose(qa[1], a.float().abs().amax(dim=1) / 448.0, rtol=1e-6)
    ref = da @ db.t()
    out = ops.gemm_fp8(qa, qb, out_dtype=dtype)
    assert relerr(out, ref) < 3 * EPS16[dtype]
    for _ in range(2):
        assert torch.equal(ops.gemm_fp8(qa, qb, out_dtype=dtype), out)
    full = a.float() @ b.float().t()
    assert relerr(out, full) < 5e-2                      # e4m3 with per-row scales: a few % of the output range
    if N % 4 == 0:
        bias, resid = rnd(N, dtype=dtype, seed=2), rnd(M, N, dtype=dtype, seed=3)
        z = ops.gemm_fp8(qa, qb, out_dtype=dtype, bias=bias, resid=resid)
        assert relerr(z, ref + bias.float() + resid.float()) < 4 * EPS16[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_fp8_fused_rope_and_swiglu_match_unfused(ops, dtype):
    """The fp8 GEMM shares the staged store phases of the 16-bit kernel: RoPE / SwiGLU fused behind the fp8 product must be
    bit-identical to the fp8 GEMM followed by the stand-alone kernels."""
    B, S, H, D, K = 2, 200, 2, 128, 256
    T, d = B * S, H * D
    x, w = rnd(T, K, dtype=dtype), rnd(3 * d, K, dtype=dtype, seed=1, scale=0.5)
    qx, qw = ops.quant_fp8_rows(x), ops.quant_fp8_rows(w)
    tab = ops.rope_table(S, D, 10000.0, dev())
    ref = ops.gemm_fp8(qx, qw, out_dtype=dtype)
    ops.rope_qk_(ref, tab, S, H, D)
    assert torch.equal(ops.gemm_fp8_rope(qx, qw, tab, S, H, D, out_dtype=dtype), ref)
    ff = 640
    wgu = rnd(2 * ff, K, dtype=dtype, seed=2, scale=0.5)
    qg = ops.quant_fp8_rows(wgu)
    gu_ref = ops.gemm_fp8(qx, qg, out_dtype=dtype)
    gu, act = ops.gemm_fp8_swiglu_fwd(qx, qg, out_dtype=dtype)
    assert torch.equal(gu, gu_ref) and torch.equal(act, ops.swiglu_fwd(gu_ref))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,S,H,causal,ragged", [(2, 256, 2, True, False), (1, 1000, 3, True, False), (2, 613, 2, True, True), (1, 1536, 2, False, False),
                                                  (3, 700, 1, False, True), (1, 4096, 2, True, False), (2, 2432, 1, True, True), (1, 64, 1, True, False),
                                                  (2, 130, 1, True, True), (1, 320, 2, False, False)])
def test_attention_forward_one_wave_per_simd_form_is_bit_identical(ops, dtype, B, S, H, causal, ragged):
    """csrc/attn_fwd4.hip (mh_attn_fwd_pingpong(2): 4 waves x 64 query rows, one wave per SIMD, K fragments resident in AccVGPRs, the softmax of
    one 32-row half placed between the MFMAs of the other) performs the per-row arithmetic of attn_fwd2 in the same order: outputs and
    log-sum-exps must agree BIT FOR BIT - causal and not, ragged lengths, sequence lengths that are not a multiple of the 256-row block,
    a single tile, padded rows zero."""
    D = 128
    g = torch.Generator(device="cuda").manual_seed(S + H)
    qkv = (torch.randn(B * S, 3 * H * D, generator=g, device="cuda") * 0.7).to(dtype)
    q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
    lens = None
    if ragged:
        lens = torch.tensor([max(1, S - 37 * (b + 1) - (b * 211) % S // 3) for b in range(B)], dtype=torch.int32, device="cuda")
    try:
        ops.attn_fwd_pingpong(0)
        o0, l0 = ops.attn_fwd2(q, k, v, B, S, H, D, causal, seqlens=lens)
        ops.attn_fwd_pingpong(2)
        o1, l1 = ops.attn_fwd2(q, k, v, B, S, H, D, causal, seqlens=lens)
        o2, _ = ops.attn_fwd2(q, k, v, B, S, H, D, causal, seqlens=lens)
    finally:
        ops.attn_fwd_pingpong(0)
    assert torch.equal(o1, o2)  # deterministic
    assert torch.isfinite(o1.float()).all()
    assert torch.equal(o1, o0), relerr(o1, o0.float())
    assert torch.equal(l1[:, :, :S], l0[:, :, :S])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,S,H,causal,ragged", [(2, 256, 2, True, False), (1, 1000, 3, True, False), (2, 613, 2, True, True), (1, 1536, 2, False, False),
                                                  (3, 700, 1, False, True), (1, 4096, 2, True, False), (2, 2432, 1, True, True)])
def test_attention_forward_pingpong_form_matches(ops, dtype, B, S, H, causal, ragged):
    """csrc/attn_fwd3.hip (mh_attn_fwd_pingpong(1): 8-wave 256-query blocks, SIMD partners in opposite phases) against the 128-query form
    (itself held to a chunked fp32 reference elsewhere in this file) and against fp32 torch on one head: ragged lengths, sequence
    lengths that are not a multiple of the 256-row block, causal and not, padded rows zero."""
    D = 128
    g = torch.Generator(device="cuda").manual_seed(S + H)
    qkv = (torch.randn(B * S, 3 * H * D, generator=g, device="cuda") * 0.7).to(dtype)
    q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
    lens = None
    if ragged:
        lens = torch.tensor([max(1, S - 37 * (b + 1) - (b * 211) % S // 3) for b in range(B)], dtype=torch.int32, device="cuda")
    try:
        ops.attn_fwd_pingpong(False)
        o0, l0 = ops.attn_fwd2(q, k, v, B, S, H, D, causal, seqlens=lens)
        ops.attn_fwd_pingpong(True)
        o1, l1 = ops.attn_fwd2(q, k, v, B, S, H, D, causal, seqlens=lens)
        o2, _ = ops.attn_fwd2(q, k, v, B, S, H, D, causal, seqlens=lens)
    finally:
        ops.attn_fwd_pingpong(False)
    assert torch.equal(o1, o2)  # deterministic
    assert relerr(o1, o0.float()) < 2 * EPS16[dtype], relerr(o1, o0.float())
    assert float((l1[:, :, :S] - l0[:, :, :S]).abs().max()) < 1e-3
    # fp32 torch on batch 0, head 0
    n = int(lens[0]) if lens is not None else S
    qf, kf, vf = (t[:n, :D].float() for t in (q, k, v))
    sc = qf @ kf.t() / D ** 0.5
    if causal:
        sc = sc.masked_fill(torch.ones(n, n, device="cuda", dtype=torch.bool).triu(1), float("-inf"))
    ref = torch.softmax(sc, -1) @ vf
    assert relerr(o1[:n, :D], ref) < 4 * EPS16[dtype]
    if n < S:
        assert float(o1[n:S, :D].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,d,ld", [(27696, 1024, 1024), (1000, 3072, 3072), (777, 1024, 3072), (77, 40, 40), (5, 4096, 4096), (48, 8 * 577, 8 * 577), (300, 264, 272)])
def test_colsum_matches_torch(ops, dtype, rows, d, ld):
    """Bias / position-embedding gradients: column sums of a (possibly strided) 16-bit matrix, fresh and accumulating; the 16-byte-per-lane
    form (d % 8 == 0, aligned rows) and the scalar fallback."""
    full = rnd(rows, ld, dtype=dtype, seed=rows + d)
    x = full[:, :d]
    ref = x.float().sum(0)
    out = torch.zeros(d, dtype=dtype, device=dev())
    ops.colsum(x, out)
    tol = 2 * EPS16[dtype] * float(x.float().abs().sum(0).max()) + 1e-6
    assert float((out.float() - ref).abs().max()) <= tol
    prev = rnd(d, dtype=dtype, seed=7)
    out2 = prev.clone()
    ops.colsum(x, out2, accumulate=True)
    assert float((out2.float() - (ref + prev.float())).abs().max()) <= tol + 2 * EPS16[dtype] * float(prev.float().abs().max())
    out3 = torch.zeros(d, dtype=dtype, device=dev())
    ops.colsum(x, out3)
    assert torch.equal(out, out3)  # deterministic


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(613, 4096, 4096), (613, 4096, 11008), (577, 1024, 4096), (40, 4096, 12288), (1000, 1032, 2048)])
def test_skinny_gemm_split_k_with_epilogues(ops, dtype, M, N, K, monkeypatch):
    """Few output tiles, long contraction: ops.gemm_nt splits K (mh_gemm_splitk_epi: fp32 partials, the epilogue applied by the fixed-order
    reduce pass).  Every epilogue and operand form against fp32 torch and against the one-pass kernel; deterministic."""
    from merlin_amd import ops as O
    assert int(O.L.lib().mh_gemm_splitk_max(M, N, K)) > 1
    a, b = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, seed=1, scale=0.5)
    bias, resid = rnd(N, dtype=dtype, seed=2), rnd(M, N, dtype=dtype, seed=3)
    bt = b.t().contiguous()
    ref = a.float() @ b.float().t()
    z = ref + bias.float()

    def run():
        r = dict(nt=ops.gemm_nt(a, b), nn=ops.gemm_nt(a, bt, b_t=True), bias=ops.gemm_nt(a, b, bias=bias),
                 gelu=ops.gemm_nt(a, b, bias=bias, act="quick_gelu"), resid=ops.gemm_nt(a, bt, b_t=True, resid=resid),
                 bias_resid=ops.gemm_nt(a, b, bias=bias, resid=resid), f32=ops.gemm_nt(a, b, out_f32=True))
        acc = rnd(M, N, dtype=dtype, seed=4)
        ops.gemm_nt(a, b, out=acc, accum=True)
        r["accum"] = acc
        return r

    monkeypatch.setattr(O, "SKINNY_SPLITK", True)
    monkeypatch.setattr(O, "SKINNY_SPLITK_ALL", True)  # (the default keeps fp16 on the one-pass order: ops._skinny_splitk_ok)
    split = run()
    again = run()
    # one grow-only workspace per (device, stream): repeated and differently sized calls on this stream share ONE buffer
    assert sum(1 for k in O._splitk_ws if k[1] == torch.cuda.current_stream().cuda_stream) == 1
    monkeypatch.setattr(O, "SKINNY_SPLITK", False)
    one = run()
    want = dict(nt=ref, nn=ref, bias=z, gelu=z * torch.sigmoid(1.702 * z), resid=ref + resid.float(), bias_resid=z + resid.float(), f32=ref,
                accum=ref + rnd(M, N, dtype=dtype, seed=4).float())
    for k, w in want.items():
        tol = 1e-5 if k == "f32" else 3 * EPS16[dtype]
        assert relerr(split[k], w) < tol, k
        assert relerr(split[k], one[k].float()) < 2 * EPS16[dtype], (k, "vs the one-pass kernel")
        assert torch.equal(split[k], again[k]), (k, "deterministic")


def test_skinny_gemm_in_a_captured_graph_is_bit_identical_to_eager(ops):
    """the split-K decision is the same inside a HIP-graph capture as outside (the partials then live in the graph's own pool), so
    a captured prefill / wide decode step sums in the eager path's order: bit-identical results, replay after replay."""
    from merlin_amd import ops as O
    M, N, K = 613, 4096, 11008
    assert int(O.L.lib().mh_gemm_splitk_max(M, N, K)) > 1 and O._skinny_splitk_ok(torch.bfloat16)
    a, b = rnd(M, K, dtype=torch.bfloat16), rnd(N, K, dtype=torch.bfloat16, seed=1, scale=0.5)
    resid = rnd(M, N, dtype=torch.bfloat16, seed=3)
    eager = (ops.gemm_nt(a, b), ops.gemm_nt(a, b, resid=resid))
    torch.cuda.synchronize()
    n_ws = len(O._splitk_ws)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            out = (ops.gemm_nt(a, b), ops.gemm_nt(a, b, resid=resid))
    torch.cuda.current_stream().wait_stream(side)
    assert len(O._splitk_ws) == n_ws  # nothing from the graph's pool was cached
    for _ in range(2):
        for o in out:
            o.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out[0], eager[0]) and torch.equal(out[1], eager[1])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,vd,vff", [(1154, 128, 256), (27696, 1024, 4096), (600, 192, 520)])
def test_gemm_with_fused_quick_gelu_is_bit_identical_to_unfused(ops, dtype, T, vd, vff):
    """CLIP MLP: fc1 + bias with quick-GELU in the store phase emitting BOTH f1 and a (mh_gemm_gelu_fwd), and the fc2 dgrad with the
    quick-GELU backward in its store phase (mh_gemm_gelu_bwd: dy W2 never stored) against mh_gemm + the stand-alone element-wise
    kernels on the stored 16-bit tensors: bit for bit (both work on the rounded tile)."""
    h2, w1, b1 = rnd(T, vd, dtype=dtype), rnd(vff, vd, dtype=dtype, seed=1, scale=0.3), rnd(vff, dtype=dtype, seed=2)
    try:
        ops.gemm_force_kernel(256)  # same kernel, same summation order for the reference
        f1_ref = ops.gemm_nt(h2, w1, bias=b1)
        a_ref = ops.quick_gelu_fwd(f1_ref)
        f1, a = ops.gemm_gelu_fwd(h2, w1, b1)
        assert torch.equal(f1, f1_ref) and torch.equal(a, a_ref)
        z = h2.float() @ w1.float().t() + b1.float()
        assert relerr(a, z * torch.sigmoid(1.702 * z)) < 4 * EPS16[dtype]
        dy, w2 = rnd(T, vd, dtype=dtype, seed=3, scale=0.5), rnd(vd, vff, dtype=dtype, seed=4, scale=0.3)
        da = ops.gemm_nt(dy, w2, b_t=True)
        df1_ref = ops.quick_gelu_bwd(f1_ref, da)
        df1 = ops.gemm_gelu_bwd(dy, w2, f1_ref)
        assert torch.equal(df1, df1_ref)
    finally:
        ops.gemm_force_kernel(0)
    f1b, ab = ops.gemm_gelu_fwd(h2, w1, b1)  # default dispatch
    assert relerr(f1b, f1_ref.float()) < 2 * EPS16[dtype] and relerr(ab, a_ref.float()) < 3 * EPS16[dtype]
    assert relerr(ops.gemm_gelu_bwd(dy, w2, f1_ref), df1_ref.float()) < 3 * EPS16[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_fp32_stream_start_kernels(ops, dtype):
    """The fp32 residual streams start from fp32 tensors (round 4): class / position embeddings added to the fp32 patch projection, pre_layrnorm
    fp32 -> fp32 (+ the 16-bit copy of its input for the backward), the projector's fp32 output spliced with widened embedding rows."""
    N, G2, d = 3, 16, 128
    S = G2 + 1
    patch32 = torch.randn(N * S, d, device=dev())
    cls, pos = rnd(d, dtype=dtype, seed=1), rnd(S, d, dtype=dtype, seed=2)
    x = ops.vit_assemble_f32(patch32, cls, pos, N, G2)
    ref = torch.cat([cls.float().expand(N, 1, d), patch32.view(N, S, d)[:, 1:]], 1) + pos.float()[None]
    assert x.dtype == torch.float32 and torch.equal(x.view(N, S, d), ref)
    w, b = rnd(d, dtype=dtype, seed=3), rnd(d, dtype=dtype, seed=4)
    y, x16 = ops.layernorm_f32_to_f32(x, w, b, 1e-5, want_x16=True)
    want = torch.nn.functional.layer_norm(x, (d,), w.float(), b.float(), 1e-5)
    assert y.dtype == torch.float32 and relerr(y, want) < 1e-5 and torch.equal(x16, x.to(dtype))
    y2, none = ops.layernorm_f32_to_f32(x, w, b, 1e-5)
    assert none is None and torch.equal(y2, y)
    # splice
    V, P = 50, 4
    ids = torch.randint(0, V, (2, 12), device=dev())
    src = torch.full((2, 12), -1, dtype=torch.int32, device=dev())
    src[0, 2:6] = torch.arange(0, 4, dtype=torch.int32)
    src[1, 5:9] = torch.arange(4, 8, dtype=torch.int32)
    emb, feats32 = rnd(V, d, dtype=dtype, seed=5), torch.randn(8, d, device=dev())
    out = ops.embed_splice_fwd_f32(ids.view(-1), src.view(-1), emb, feats32)
    want = emb.float()[ids.view(-1)].clone()
    sel = src.view(-1) >= 0
    want[sel] = feats32[src.view(-1)[sel].long()]
    assert out.dtype == torch.float32 and torch.equal(out, want)
    assert torch.equal(ops.embed_splice_fwd_f32(ids.view(-1), None, emb, None), emb.float()[ids.view(-1)])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("d,ln", [(4096, False), (1024, True), (4096, True), (1536, False), (1536, True)])
def test_fp32_stream_reader_widths(ops, dtype, d, ln):
    """The fp32 streams' reader (mh_norm_fwd_f32in) at the model's two widths - the register-resident forms, which request a row once - and
    at a width that takes the generic three-pass loop: against torch in fp32; the 16-bit copy of the input is the rounded input exactly."""
    rows = 37
    x = torch.randn(rows, d, device=dev()) * 3 + 0.5
    w = rnd(d, dtype=dtype, seed=1, scale=1.0)
    b = rnd(d, dtype=dtype, seed=2, scale=1.0) if ln else None
    y, x16 = ops.norm_fwd_f32in(x, w, 1e-5, b=b, want_x16=True)
    if ln:
        want = torch.nn.functional.layer_norm(x, (d,), w.float(), b.float(), 1e-5)
    else:
        want = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * w.float()
    assert y.dtype == dtype and relerr(y, want) < 2 * EPS16[dtype]
    assert torch.equal(x16, x.to(dtype))
    y2, none = ops.norm_fwd_f32in(x, w, 1e-5, b=b)
    assert none is None and torch.equal(y2, y)
