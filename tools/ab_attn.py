import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, ctypes
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "dev_arms")); import dev_ops as D  # needs MH_LIB_PATH=tools/dev_arms/libmerlin_hip_dev.so (python -m merlin_amd.csrc.build --dev)
from merlin_amd import ops as O, _lib as L
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import bench_ops as B
for split in (1, 0):
    L.lib().mh_attn_bwd_split(ctypes.c_int(split)); print("split", split)
    B.bench_attn(8, 4096, 32, 128, True)
L.lib().mh_attn_bwd_split(ctypes.c_int(1))
B.bench_attn(48, 577, 16, 64, False)
