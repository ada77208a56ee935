"""The logits error of the HIP path held against a MEASURED floor.

BASELINE's tolerance - logits within 1e-3 (relative to max|logit|) of the reference's fp32 CPU path - is stated for fp16.  A matrix
unit that takes 16-bit operands cannot meet it at the 7B model's full depth whatever the kernels do: the floor below is the SAME fp32
CPU oracle with nothing changed except that every matrix-unit operand (activation side of every Linear, rotated q / k, v, softmax
probabilities) is rounded to the 16-bit type, and - for the 16-bit-stream rows - the residual streams too (oracle.ref_cpu.rounding).
Accumulation, norms, softmax, RoPE and SwiGLU stay fp32 there.  Its error against the reference golden is what ANY single-pass
16-bit-operand implementation pays ("operand-only floor").  The HIP path additionally keeps its GEMM outputs - q | k before the rotation,
gate | up before SwiGLU - in 16 bits (the fused epilogues work on the rounded values, bit-identical to the unfused kernels): the same
oracle with those roundings added (rounding(outputs=True)) is the "storage-model floor" = what the path's DATA FORMAT costs whatever the
kernels do.  This file asserts, for the rms, the 99.9th percentile and the single-element maximum, on the medium model and at full 7B depth, that the
engine's DEFAULT configuration (fp32 residual streams that start from fp32 tensors; RoPE / SwiGLU on the fp32 accumulators of the 4-wave GEMM) stays
within a stated factor of the OPERAND-ONLY floor, and the 16-bit-stream configuration within the same factor of its storage-model floor.
The 1e-3 tolerance itself is met at every size by the fp32-store parity mode (tests/test_parity_mode_gpu.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")

# factors the HIP statistics are held to (measured: profiles/r04_parity_floor.txt).  Default configuration (fp32 residual streams, which since
# round 4 also START from fp32 tensors, RoPE / SwiGLU on the fp32 accumulators): against the OPERAND-ONLY floor itself - rms 1.10 (measured 1.00 - 1.02),
# p99.9 1.15 (0.95 - 1.05), single-element maximum 1.35 (one element of 40 k: 0.82 - 1.15).  16-bit streams: the same factors against the storage-model floor
# (operands + the 16-bit streams and the 16-bit tensors at their start), rms 1.25 against the operand-only floor (measured 1.01 - 1.10).
F_RMS, F_P999, F_MAX, F_RMS_OPERANDS_16 = 1.10, 1.15, 1.35, 1.25


def _stats(got, g):
    d = np.abs(got.astype(np.float64) - g["logits_slice"].astype(np.float64)) / float(g["logits_absmax"])
    return float(d.max()), float(np.quantile(d, 0.999)), float(np.sqrt((d ** 2).mean()))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name", ["medium_cfg1", "full_cfg1"])
def test_hip_logits_error_within_a_factor_of_the_16bit_operand_floor(name, dtype):
    import psutil

    from oracle import cases as C
    from oracle import ref_cpu as R
    from test_model_gpu import _build, _to_dev

    if name == "full_cfg1":
        assert psutil.virtual_memory().available >= 60e9, "the fp32 CPU oracle of the 7B model needs ~40 GB of free host memory"
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    cfg, batch = C.get_case(name)
    g = np.load(os.path.join(GOLD, name + ".npz"))
    sl = (slice(None), slice(None, None, 16), slice(0, 256)) if name == "full_cfg1" else (slice(None), slice(None, None, 8), slice(0, 512))
    model = _build(cfg, dtype)
    assert model.engine.fp32_residual is True  # the default configuration IS the one whose parity is quoted
    hip = {}
    with torch.no_grad():
        for stream32 in (True, False):
            model.engine.fp32_residual = stream32
            hip[stream32] = _stats(model(**_to_dev(batch)).logits.float()[sl].cpu().numpy(), g)
        P = {k: p.detach().float().cpu() for k, p in model.named_parameters()}  # the generator's bits (checked in test_model_gpu)
        del model
        torch.cuda.empty_cache()
        floor, floor_op = {}, {}
        for stream32 in (True, False):
            for outputs, dst in ((True, floor), (False, floor_op)):
                with R.rounding(dtype, stream=not stream32, outputs=outputs):
                    _, lg = R.forward(P, cfg, batch["input_ids"], batch["attention_mask"], batch["labels"], batch["images"])
                dst[stream32] = _stats(lg[sl].numpy(), g)
    bad = []
    for stream32 in (True, False):
        (hm, hp, hr), (fm, fp_, fr), (om, op_, or_) = hip[stream32], floor[stream32], floor_op[stream32]
        print(f"[floor {name} {str(dtype).split('.')[-1]} {'fp32' if stream32 else '16-bit'} stream] "
              f"HIP max {hm:.3e} p99.9 {hp:.3e} rms {hr:.3e} | storage-model floor max {fm:.3e} p99.9 {fp_:.3e} rms {fr:.3e} "
              f"(ratios {hm / fm:.2f} {hp / fp_:.2f} {hr / fr:.2f}) | operand-only floor max {om:.3e} p99.9 {op_:.3e} rms {or_:.3e} "
              f"(ratios {hm / om:.2f} {hp / op_:.2f} {hr / or_:.2f})")
        if stream32:  # the default: held to the operand-only floor itself
            fail = hr > F_RMS * or_ or hp > F_P999 * op_ or hm > F_MAX * om
        else:
            fail = hr > F_RMS * fr or hp > F_P999 * fp_ or hm > F_MAX * fm or hr > F_RMS_OPERANDS_16 * or_
        if fail:
            bad.append((stream32, hip[stream32], floor[stream32], floor_op[stream32]))
    assert not bad, bad
    # and the fp32 stream is never worse than the 16-bit one (rms)
    assert hip[True][2] <= hip[False][2] * 1.02
