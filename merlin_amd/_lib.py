"""ctypes binding of libmerlin_hip.so (include/merlin_hip.h).  Fails loudly when the library is
missing: there is NO CPU / eager fallback for the product path."""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# MH_LIB_PATH: kernel-development override (tools/dev_arms/libmerlin_hip_dev.so = product kernels + A/B arms)
LIB_PATH = os.environ.get("MH_LIB_PATH") or os.path.join(_HERE, "csrc", "libmerlin_hip.so")
HEADER = os.path.join(_HERE, "..", "include", "merlin_hip.h")

MH_BF16, MH_F16, MH_F32 = 0, 1, 2
EPI_BIAS, EPI_QUICK_GELU, EPI_RESIDUAL, EPI_ACCUM, EPI_OUT_F32 = 1, 2, 4, 8, 16

_lib = None


class MerlinHipError(RuntimeError):
    pass


def declared_symbols() -> list:
    """Every function name declared in include/merlin_hip.h."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mh_[a-z0-9_]+)\s*\(", txt)))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MerlinHipError(
                f"{LIB_PATH} not found: build it with `python -m merlin_amd.csrc.build` "
                "(hipcc --offload-arch=gfx950).  merlin_amd has no CPU fallback.")
        # torch must load ITS HIP runtime first: if libmerlin_hip.so pulled in a second copy of libamdhip64
        # before torch, the process would hold two runtimes and this one would see no device.
        import torch  # noqa: F401

        _lib = C.CDLL(LIB_PATH)
        _lib.mh_strerror.restype = C.c_char_p
        _lib.mh_strerror.argtypes = [C.c_int]
        if hasattr(_lib, "mh_attn_bwd_ws_elems"):  # dev library only
            _lib.mh_attn_bwd_ws_elems.restype = C.c_int64
        if os.environ.get("MH_GEMM_PERSISTENT") == "0":  # A/B switches for benchmarks
            _lib.mh_gemm_persistent(C.c_int(0))
        if os.environ.get("MH_ATTN_BWD_FUSED_KV"):  # 0: dK, dV from two kernels; 1: attn_bwd2_kv_k<MODE 3>; 2: attn_bwd3_kv_k (default)
            _lib.mh_attn_bwd_fused_kv(C.c_int(int(os.environ["MH_ATTN_BWD_FUSED_KV"])))
        if os.environ.get("MH_ATTN_WIDE_STORES"):
            _lib.mh_attn_wide_stores(C.c_int(int(os.environ["MH_ATTN_WIDE_STORES"])))
        if os.environ.get("MH_W4_MASK"):  # layouts the auto selection gives to the 4-wave GEMM (bit 0 TN, 1 NN, 2 NT)
            _lib.mh_gemm_w4_policy(C.c_int(int(os.environ["MH_W4_MASK"])))
        if os.environ.get("MH_W4_HALF"):  # 128-row block tiles of the 4-wave GEMM: 0 never, 1 auto (default), 2 wherever the form exists
            _lib.mh_gemm_w4_half(C.c_int(int(os.environ["MH_W4_HALF"])))
        if os.environ.get("MH_GEMM_GM"):
            _lib.mh_gemm_raster_group(C.c_int(int(os.environ["MH_GEMM_GM"])))
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib().mh_strerror(code).decode()
        raise MerlinHipError(f"{what} failed: {msg} (code {code})")


def p(t):
    """device pointer of a tensor (or None)."""
    return C.c_void_p(0 if t is None else t.data_ptr())


i32, i64, f32, u64 = C.c_int, C.c_int64, C.c_float, C.c_uint64
