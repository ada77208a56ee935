R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r04_gputests_4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_gputests_4.log
tail -5 gpurun_out/r04_gputests_4.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/bench_r04_gelu.json 2> gpurun_out/bench_r04_gelu.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_r04_gelu.json")); print(d["ms_per_step"], d["roofline"]["frac"], d["forward_only"]["ms_per_step"], d["config"]["loss"])
PY
