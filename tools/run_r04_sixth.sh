R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r04_gputests_5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_gputests_5.log
tail -4 gpurun_out/r04_gputests_5.log
grep -h "floor \|full cfg1\|FAILED\|passed" gpurun_out/r04_gputests_5.log | sed 's/^\.*//' | cut -c1-330
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/bench_r04_f32start.json 2> gpurun_out/bench_r04_f32start.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_r04_f32start.json")); print(d["ms_per_step"], d["roofline"]["frac"], d["forward_only"]["ms_per_step"], d["config"]["loss"], d["peak_hbm_gb"])
PY
