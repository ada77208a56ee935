"""Build libmerlin_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m merlin_amd.csrc.build [--force]

One object per .hip file (parallel), linked into merlin_amd/csrc/libmerlin_hip.so (in-tree so
it travels to the GPU box with the snapshot).  No torch dependency: the ABI is plain C.

`--dev` additionally builds tools/dev_arms/libmerlin_hip_dev.so: the same sources compiled with -DMH_DEV_ARMS plus the
non-dispatched A/B arms under tools/dev_arms/ (alternative GEMM tilings, first-generation attention kernels).  The product
library never contains them; kernel-development tools select the dev library with MH_LIB_PATH.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["gemm.hip", "gemm256.hip", "gemm_w4.hip", "norm.hip", "elementwise.hip", "attn_fwd2.hip", "attn_fwd3.hip", "attn_fwd4.hip", "attn_bwd2.hip", "loss_splice.hip", "conv.hip", "decode.hip", "sampling.hip", "fp8_quant.hip", "parity32.hip"]
DEV_DIR = os.path.normpath(os.path.join(HERE, "..", "..", "tools", "dev_arms"))
DEV_SOURCES = ["gemm256_m32.hip", "gemm256w4.hip", "gemm256w8.hip", "attn_fwd.hip", "attn_bwd.hip", "attn_bwd_kv.hip"]
DEV_LIB = os.path.join(DEV_DIR, "libmerlin_hip_dev.so")
LIB = os.path.join(HERE, "libmerlin_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result"]


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.exists(c) or c == "hipcc":
            return c


def _newer(src, dst):
    if not os.path.exists(dst):
        return True
    deps = [src] + [os.path.join(HERE, h) for h in os.listdir(HERE) if h.endswith(".h")] + [os.path.join(HERE, "..", "..", "include", "merlin_hip.h")]
    return any(os.path.getmtime(d) > os.path.getmtime(dst) for d in deps)


def _compile(src, force, srcdir=HERE, objdir=None, extra=()):
    obj = os.path.join(objdir or os.path.join(HERE, "build"), src.replace(".hip", ".o"))
    os.makedirs(os.path.dirname(obj), exist_ok=True)
    if not force and not _newer(os.path.join(srcdir, src), obj):
        return obj
    cmd = [_hipcc(), *FLAGS, *extra, "-I", HERE, "-c", os.path.join(srcdir, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[merlin_amd] built {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB)")
    return LIB


def build_dev(force: bool = False, verbose: bool = True) -> str:
    """Product sources with -DMH_DEV_ARMS + the A/B arms -> tools/dev_arms/libmerlin_hip_dev.so (kernel development only)."""
    objdir = os.path.join(DEV_DIR, "build")
    jobs = [(s, HERE) for s in SOURCES] + [(s, DEV_DIR) for s in DEV_SOURCES]
    with cf.ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(lambda j: _compile(j[0], force, srcdir=j[1], objdir=objdir, extra=("-DMH_DEV_ARMS",)), jobs))
    r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", DEV_LIB], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[merlin_amd] built {DEV_LIB} ({os.path.getsize(DEV_LIB) / 1e6:.1f} MB)")
    return DEV_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    if "--dev" in sys.argv:
        build_dev(force="--force" in sys.argv)
