R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fp8_training_gpu.py -q -p no:cacheprovider 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt5
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt5 -o r -- python $R/bench.py --config cfg5 --steps 3 --warmup 3 --no-cpu-baseline --no-forward-leg > /tmp/kt5.log 2>&1
f5=$(find /tmp/kt5 -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $f5 > $R/gpurun_out/r04_step_cfg5_fp8_kernel_stats_v5.txt 2>&1
grep "quant\|absmax" $R/gpurun_out/r04_step_cfg5_fp8_kernel_stats_v5.txt | cut -c1-60,112-170
cd $R; python bench.py --config cfg5 --steps 6 --warmup 3 --no-cpu-baseline --no-forward-leg > gpurun_out/bench_r04_cfg5_v5.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/bench_r04_cfg5_v5.json')); print('cfg5', d['ms_per_step'], d['value'], d['config']['loss'])"
