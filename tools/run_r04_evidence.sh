# Round-4 evidence bundle (run on the GPU box from the repo root; LAST GPU action after any change to the GEMM sources): rocprofv3 kernel
# stats of the default bench command and of cfg 5, the PMC traffic passes (stamped with the GEMM source hash), the N>1 plumbing on one GPU
# with the persistent 8-wave launch on and off, and the default bench line.  Outputs under gpurun_out/ (copied into profiles/ by hand).
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-v1}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-extras --no-forward-leg > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $f > $R/gpurun_out/r04_step_cfg3_kernel_stats_$TAG.txt 2>&1 || ls -R /tmp/kt | head
tail -1 /tmp/kt.log > $R/gpurun_out/r04_step_cfg3_bench_under_rocprof_$TAG.json
rm -rf /tmp/kt5
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt5 -o r -- python $R/bench.py --config cfg5 --steps 3 --warmup 3 --no-cpu-baseline --no-forward-leg > /tmp/kt5.log 2>&1
f5=$(find /tmp/kt5 -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $f5 > $R/gpurun_out/r04_step_cfg5_fp8_kernel_stats_$TAG.txt 2>&1 || ls -R /tmp/kt5 | head
tail -1 /tmp/kt5.log > $R/gpurun_out/r04_step_cfg5_bench_under_rocprof_$TAG.json
if [ "${SKIP_PMC:-0}" != "1" ]; then bash $R/tools/pmc_step_traffic.sh r04_gemm_traffic.json > /dev/null 2>&1; fi
cd $R
python bench.py --steps 4 --warmup 2 --force-dp --no-cpu-baseline --no-extras > gpurun_out/bench_r04_force_dp_$TAG.json 2> gpurun_out/bench_r04_force_dp_$TAG.err
MH_GEMM_PERSISTENT=0 python bench.py --steps 4 --warmup 2 --force-dp --no-cpu-baseline --no-extras > gpurun_out/bench_r04_force_dp_nonpersistent_$TAG.json 2>> gpurun_out/bench_r04_force_dp_$TAG.err
python bench.py --config cfg5 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r04_cfg5_$TAG.json 2> /dev/null
python tools/bench_gemm_shapes.py > gpurun_out/r04_gemm_shapes_$TAG.txt 2>&1
(cd tools && python yardstick.py > ../gpurun_out/r04_yardstick_$TAG.txt 2>&1)
if [ "${SKIP_PMC:-0}" != "1" ]; then cp gpurun_out/r04_gemm_traffic.json profiles/r04_gemm_traffic.json; fi
python bench.py --steps 10 --warmup 3 > gpurun_out/bench_r04_default_$TAG.json 2> /dev/null
head -30 gpurun_out/r04_step_cfg3_kernel_stats_$TAG.txt; head -c 1800 gpurun_out/r04_gemm_traffic.json; python -c "
import json
for n in ('default', 'cfg5', 'force_dp', 'force_dp_nonpersistent'):
    try:
        d = json.load(open('gpurun_out/bench_r04_%s_$TAG.json' % n)); print(n, d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('traffic'), d.get('comm_ms_exposed'))
    except Exception as e:
        print(n, 'failed', e)"
