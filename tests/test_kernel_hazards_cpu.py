"""csrc/attn_fwd4.hip issues its MFMAs from inline asm, so hipcc inserts none of the wait states an MFMA needs and is free to place register
copies directly in front of / behind them (profiles/r04_attn_fwd_wave64.txt, hazards 1-3: silent wrong results, found on hardware).  What
keeps the kernel correct is where the compiler put things in THIS build, so the listing itself is checked: no spill (a spilled register
whose LDS read is in flight is garbage), and tools/check_mfma_hazards.py finds no instruction that reads an MFMA result too early and no
MFMA that reads a register written less than two wait states before it.  hipcc cross-compiles gfx950 without a GPU."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "merlin_amd", "csrc")


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    return None


@pytest.mark.skipif(_hipcc() is None, reason="hipcc not installed")
def test_attn_fwd4_listing_has_no_spills_and_no_mfma_hazards(tmp_path):
    from merlin_amd.csrc import build

    out = tmp_path / "attn_fwd4.s"
    cmd = [_hipcc(), *build.FLAGS, "-I", CSRC, "--cuda-device-only", "-S", os.path.join(CSRC, "attn_fwd4.hip"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text()
    kernels = re.findall(r"^(_ZN6mhattn\S*attn_fwd4_k\S*):", text, flags=re.M)
    assert len(set(kernels)) == 4, kernels  # bf16 / fp16 x causal / full
    spills = [int(x) for x in re.findall(r"\.vgpr_spill_count:\s*(\d+)", text)]
    assert spills and max(spills) == 0, spills
    assert "scratch_load" not in text and "scratch_store" not in text
    listing = "\n".join(line for line in text.split("\n") if "sched_barrier" not in line and "ASMSTART" not in line and "ASMEND" not in line)
    clean = tmp_path / "attn_fwd4_clean.s"
    clean.write_text(listing)
    for k in sorted(set(kernels)):
        c = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_mfma_hazards.py"), str(clean), k], capture_output=True, text=True)
        assert c.returncode == 0, c.stderr
        assert c.stdout.strip().endswith("hazards: 0"), (k, c.stdout[-1500:])


def test_hazard_checker_sees_a_stale_read_and_a_late_write(tmp_path):
    """The checker itself: an accumulator copied directly behind its MFMA, and an AccVGPR written directly in front of the MFMA that reads it."""
    chk = os.path.join(ROOT, "tools", "check_mfma_hazards.py")
    bad = tmp_path / "bad.s"
    bad.write_text("_Zk:\n\tv_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n\tv_accvgpr_read_b32 v9, a3\n"
                   "\tv_accvgpr_write_b32 a20, v1\n\tv_mfma_f32_32x32x16_bf16 v[16:31], a[20:23], v[4:7], 0\n\ts_endpgm\n")
    c = subprocess.run([sys.executable, chk, str(bad), "_Zk"], capture_output=True, text=True)
    assert c.stdout.strip().endswith("hazards: 2"), c.stdout
    good = tmp_path / "good.s"
    good.write_text("_Zk:\n\tv_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]\n\ts_nop 15\n\ts_nop 3\n\tv_accvgpr_read_b32 v9, a3\n"
                    "\tv_accvgpr_write_b32 a20, v1\n\ts_nop 1\n\tv_mfma_f32_32x32x16_bf16 v[16:31], a[20:23], v[4:7], 0\n\ts_endpgm\n")
    c = subprocess.run([sys.executable, chk, str(good), "_Zk"], capture_output=True, text=True)
    assert c.stdout.strip().endswith("hazards: 0"), c.stdout
    c = subprocess.run([sys.executable, chk, str(good), "_Znope"], capture_output=True, text=True)
    assert c.returncode == 2
