"""The 16-bit-operand floor of the logits error: the fp32 oracle with every matrix-unit operand (and, optionally, the
residual streams) rounded to 16 bits (oracle.ref_cpu.rounding), against the REFERENCE's fp32 golden logits.  CPU only.
    python tools/measure_floor.py [medium_cfg1|full_cfg1 ...]
Prints max / p99.9 / rms of |d logits| / max|ref| per (dtype, stream) - the numbers the HIP path's own error is held against in
tests/test_parity_floor_gpu.py."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cases as C  # noqa: E402
from oracle import ref_cpu as R  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def stats(got, g):
    d = np.abs(got.astype(np.float64) - g["logits_slice"].astype(np.float64)) / float(g["logits_absmax"])
    return float(d.max()), float(np.quantile(d, 0.999)), float(np.sqrt((d ** 2).mean()))


def main():
    names = sys.argv[1:] or ["medium_cfg1", "full_cfg1"]
    for name in names:
        cfg, batch = C.get_case(name)
        g = np.load(os.path.join(GOLD, name + ".npz"))
        P = R.make_params(cfg, seed=0)
        sl = (slice(None), slice(None, None, 16), slice(0, 256)) if name.startswith("full") else (slice(None), slice(None, None, 8), slice(0, 512))
        for dt in (torch.float16, torch.bfloat16):
            for stream in (True, False):
                t0 = time.time()
                with torch.no_grad(), R.rounding(dt, stream=stream):
                    _, lg = R.forward(P, cfg, batch["input_ids"], batch["attention_mask"], batch["labels"], batch["images"])
                mx, p999, rms = stats(lg[sl].numpy(), g)
                print(f"{name} {str(dt).split('.')[-1]:9s} stream={'16-bit' if stream else 'fp32  '} floor: max {mx:.3e} p99.9 {p999:.3e} rms {rms:.3e}   ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
