#!/bin/bash
# round 6, GPU call 3: the whole -m gpu suite + smoke on this tree, then the round-6 evidence bundle (tools/run_r06_evidence.sh)
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r06ev; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" >> $O/pytest_gpu.txt 2>&1; tail -1 $O/pytest_gpu.txt
bash tools/run_r06_evidence.sh > $O/evidence.log 2>&1
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r06ev/bench_r06_default.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("ms_per_step", "value", "roofline", "executed_tflops_per_gpu")})
    print(d["forward_only"]); print(d["extras"]["cfg2"]["forward_ms"], d["extras"]["cfg2"]["train_ms_per_step"], d["extras"]["cfg5"].get("ms_per_step"))
except Exception as e:
    print("bench line:", e)
PY
