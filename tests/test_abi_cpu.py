"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol
that include/merlin_hip.h declares (no compute calls: there is no GPU here)."""
import os

import pytest


def test_library_exports_every_declared_symbol():
    from merlin_amd import _lib as L

    if not os.path.exists(L.LIB_PATH):
        from merlin_amd.csrc import build

        build.build(verbose=False)
    lib = L.lib()
    names = L.declared_symbols()
    assert len(names) >= 30, names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/merlin_hip.h but not exported: {missing}"
    assert lib.mh_version() >= 100
    assert b"dtype" in lib.mh_strerror(-2)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from merlin_amd import _lib as L

    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(L.MerlinHipError):
        L.lib()


def test_product_never_imports_oracle():
    """The HIP path must not route through the CPU oracle (or any CPU fallback)."""
    import pathlib
    import re

    root = pathlib.Path(__file__).resolve().parents[1] / "merlin_amd"
    bad = []
    for f in root.rglob("*.py"):
        txt = f.read_text()
        if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M):
            bad.append(str(f))
    assert not bad, bad


def test_host_side_launch_plans():
    """Host-only entry points (no launch): the split counts mh_gemm_splitk_max hands the skinny-GEMM path of ops.gemm_nt and the
    split-KV count of the decode attention."""
    import ctypes as C

    from merlin_amd import _lib as L

    lib = L.lib()
    sk = lambda M, N, K: int(lib.mh_gemm_splitk_max(C.c_int(M), C.c_int(N), C.c_int(K)))  # noqa: E731
    assert sk(613, 4096, 4096) == 5 and sk(613, 4096, 11008) == 5   # o / down projections of one 613-token sequence: 48 tiles
    assert sk(613, 12288, 4096) == 1 and sk(613, 22016, 4096) == 1   # q|k|v and gate|up fill the chip on their own
    assert sk(32768, 4096, 4096) == 1                                # the headline shapes never split
    assert sk(613, 4096, 512) == 1                                   # short contractions are not worth the reduce pass
    for M, N, K in ((577, 1024, 4096), (40, 4096, 12288), (1000, 1032, 2048)):
        s = sk(M, N, K)
        nk = (K + 63) // 64
        assert 1 < s <= 16 and (s - 1) * ((nk + s - 1) // s) < nk, (M, N, K, s)  # no empty split
    assert int(lib.mh_attn_decode_splits(C.c_int(1), C.c_int(32), C.c_int(4096))) == 32
    assert int(lib.mh_attn_decode_splits(C.c_int(8), C.c_int(32), C.c_int(4096))) == 4


def test_grouped_wgrad_host_plan_and_struct_layout():
    """Host side of mh_wgrad_grouped: the launch-plan predicate (a CLIP-L layer's four weight gradients over 27 696 tokens are grouped, the tiny
    test models' are not, odd widths never) and the ctypes mirror of `MhWgradProblem` (include/merlin_hip.h) - 3 pointers + 3 strides + 4 ints."""
    import ctypes as C

    from merlin_amd import ops as O

    vd, vff = 1024, 4096
    assert O.wgrad_group_pays(48 * 577, [(vd, vff), (vff, vd), (vd, vd), (3 * vd, vd)])
    assert not O.wgrad_group_pays(3 * 17, [(128, 256), (256, 128), (128, 128), (384, 128)])       # tiny fixture: a few tiles, a few tokens
    assert not O.wgrad_group_pays(48 * 577, [(vd, vff), (vff, vd), (vd, vd), (3 * vd, vd + 4)])  # a width the kernel cannot take
    assert not O.wgrad_group_pays(48 * 577, [(vd, vff)] * 9)                                      # more problems than one launch holds
    assert C.sizeof(O._WgradProblem) == 64
    assert [f[0] for f in O._WgradProblem._fields_] == ["dy", "lddy", "x", "ldx", "out", "ldo", "M", "N", "accumulate", "reserved"]
