# Round-4 second GPU call: new 4-wave GEMM forms (tests), the floor / parity lines, and the dispatch-policy A/B of the cfg-3 step.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_w4_gpu.py -q -p no:cacheprovider > gpurun_out/r04_w4_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r04_w4_tests.log
tail -25 gpurun_out/r04_w4_tests.log
timeout 1800 python -m pytest tests -m gpu -q -s -p no:cacheprovider --deselect tests/test_gemm_w4_gpu.py > gpurun_out/r04_gputests_2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_gputests_2.log
tail -8 gpurun_out/r04_gputests_2.log
grep -h "^\[" gpurun_out/r04_gputests_2.log | grep -i "floor\|full cfg1\|lm_head\|fp8 train cfg 5" > gpurun_out/r04_parity_lines_2.txt
cat gpurun_out/r04_parity_lines_2.txt
for cfg in "0:11" "1:11" "1:27" "1:59" "1:123"; do
  G=${cfg%%:*}; M=${cfg##*:}
  MH_WGRAD_GROUPED=$G MH_W4_MASK=$M python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --no-forward-leg > gpurun_out/bench_r04_policy_g${G}_m${M}.json 2>> gpurun_out/bench_r04_policy.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/bench_r04_policy_g${G}_m${M}.json"))
    print("grouped=$G mask=$M", d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["config"]["loss"])
except Exception as e:
    print("grouped=$G mask=$M failed", e)
PY
done
