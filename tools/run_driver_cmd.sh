#!/bin/bash
# the driver's own command on this box: python bench.py -> gpurun_out/spread_r06/bench_$1.json + a one-line summary
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/spread_r06
python bench.py > gpurun_out/spread_r06/bench_$1.json 2> gpurun_out/spread_r06/bench_$1.err
python - "$1" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/spread_r06/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
r, e = d["roofline"], d["extras"]
print(f"step {d['ms_per_step']} ms = {d['value']} tok/s | GEMM frac {r['frac']} traffic {r['traffic']} probe {r['sustained_mfma_probe']['tflops']} TF | fwd {d['forward_only']['ms_per_step']} | "
      f"cfg2 {e['cfg2']['train_ms_per_step']} / {e['cfg2']['forward_ms']} | cfg5 {e['cfg5']['ms_per_step']} traffic {e['cfg5']['roofline']['traffic']} | cpu {d['cpu_baseline']['value']}")
PY
