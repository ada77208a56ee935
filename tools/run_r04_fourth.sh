# Round-4 GPU call: full GPU suite with the chosen dispatch defaults, default bench (with the cfg-2 extras) next to the round-3 dispatch.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r04_gputests_3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_gputests_3.log
tail -6 gpurun_out/r04_gputests_3.log
grep -h "^\[" gpurun_out/r04_gputests_3.log | grep -i "floor\|full cfg1\|lm_head\|fp8 train cfg 5" > gpurun_out/r04_parity_lines_3.txt
cat gpurun_out/r04_parity_lines_3.txt | cut -c1-330
for M in default 11; do
  if [ $M = default ]; then unset MH_W4_MASK; else export MH_W4_MASK=$M; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r04_mask_$M.json 2>> gpurun_out/bench_r04_mask.err
  python - <<PY
import json
d = json.load(open("gpurun_out/bench_r04_mask_$M.json"))
print("mask=$M", d["ms_per_step"], d["roofline"]["achieved"], d["forward_only"]["ms_per_step"], d.get("extras"))
PY
done
