#!/bin/bash
# round 6, GPU call 1: the new evidence tests, the driver's command with the rotated-batch default + measured S=4096 CPU baseline,
# and the same-box A/B of one-batch vs rotated batches (profiles/r06_batch_rotation_ab.txt)
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r06c1; mkdir -p $O
timeout 1500 python -m pytest tests/test_headline_gpu.py "tests/test_model_gpu.py::test_scored_rows_count_survives_many_outstanding_forwards" \
  "tests/test_model_gpu.py::test_lm_head_backward_over_the_scored_rows_only" "tests/test_dropin_gpu.py::test_released_geometry_448px_conv_stride2_vs_reference_golden" \
  "tests/test_geometry_gpu.py::test_packed_sequence_at_benchmark_length_vs_oracle" -x -q -s -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -5 $O/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
for arm in same rot same2 rot2; do
  flag=""; case $arm in same*) flag="--same-batch";; esac
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-forward-leg $flag > $O/bench_$arm.json 2> $O/bench_$arm.err
done
python - <<'PY'
import json
O = "gpurun_out/r06c1"
def rd(n):
    try: return json.loads(open(f"{O}/bench_{n}.json").read().strip().splitlines()[-1])
    except Exception as e: return None
with open(f"{O}/r06_batch_rotation_ab.txt", "w") as f:
    f.write("same box, back to back: python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-forward-leg [--same-batch]\n")
    for n in ("default", "same", "rot", "same2", "rot2"):
        d = rd(n)
        if d: f.write(f"{n:8s} ms/step {d['ms_per_step']:8.2f} tok/s {d['value']:9.1f} GEMM frac {d['roofline']['frac']:.4f} loss {d['config']['loss_first_warmup_step']} -> {d['config']['loss']} batches: {d['config']['batches']}\n")
print(open(f"{O}/r06_batch_rotation_ab.txt").read())
d = rd("default")
if d: print(json.dumps({k: d[k] for k in ("ms_per_step","useful_tflops_per_gpu","executed_tflops_per_gpu","mfma_roofline_frac_step","cpu_baseline")}, indent=1)[:3000]); print(d["extras"]["cfg2"]["forward_ms"], d["extras"]["cfg2"]["train_ms_per_step"], d["extras"]["cfg5"]["ms_per_step"] if isinstance(d["extras"]["cfg5"], dict) else d["extras"]["cfg5"])
PY
