R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
python tools/ab_w4_forms.py > gpurun_out/r04_w4_forms_ab.txt 2>&1; cat gpurun_out/r04_w4_forms_ab.txt
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-extras --no-forward-leg > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $f > $R/gpurun_out/r04_step_cfg3_kernel_stats_allnew.txt 2>&1
head -24 $R/gpurun_out/r04_step_cfg3_kernel_stats_allnew.txt | cut -c1-180
