# Round-4 first GPU call: the full GPU test suite, the default bench line (fp32 residual streams) next to the 16-bit-stream variant,
# and the dQ-atomics probe.  Outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
python -m merlin_amd.csrc.build > gpurun_out/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider > gpurun_out/r04_gputests_1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_gputests_1.log
tail -5 gpurun_out/r04_gputests_1.log
grep -h "^\[" gpurun_out/r04_gputests_1.log | grep -i "floor\|full cfg1\|lm_head\|fp8 train cfg 5" > gpurun_out/r04_parity_lines_1.txt
hipcc --offload-arch=gfx950 -O3 tools/probes/atomic_dq_probe.hip -o /tmp/atomic_dq_probe > /dev/null 2>&1 && /tmp/atomic_dq_probe > gpurun_out/r04_atomic_dq_probe.txt 2>&1
cat gpurun_out/r04_atomic_dq_probe.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r04_start_fp32stream.json 2> gpurun_out/bench_r04_start.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --residual-16bit > gpurun_out/bench_r04_start_16bitstream.json 2>> gpurun_out/bench_r04_start.err
python - <<'PY'
import json
for f in ("bench_r04_start_fp32stream", "bench_r04_start_16bitstream"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, d["ms_per_step"], d["value"], d["roofline"]["frac"], d.get("forward_only", {}).get("ms_per_step"), d.get("extras"))
    except Exception as e:
        print(f, "failed", e)
PY
