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
