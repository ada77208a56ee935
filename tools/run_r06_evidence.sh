#!/bin/bash
# Round-6 evidence bundle (GPU box, from the repo root; LAST GPU action after any kernel change).  Everything lands in gpurun_out/r06ev/ and is
# copied into profiles/ by hand (profiles/README.md says which file backs which claim).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06ev; mkdir -p $O; cd $R
# 1. the driver's own command
timeout 900 python bench.py > $O/bench_r06_default.json 2> $O/bench_r06_default.err
# 2. rocprofv3 --kernel-trace --stats of the cfg-3 step and of the cfg-5 fp8 step (6 steps each)
tools/run_prof.sh r06ev/r06_step_cfg3_kernel_stats python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-extras --no-cfg5-extra --no-forward-leg
tools/run_prof.sh r06ev/r06_step_cfg5_fp8_kernel_stats python $R/bench.py --config cfg5 --steps 3 --warmup 3 --no-cpu-baseline --no-forward-leg --no-cfg5-extra
# 3. cfg 2 (S = 613) forward and training step
MODE=fwd tools/run_prof.sh r06ev/r06_cfg2_fwd_kernel_stats python $R/tools/prof_cfg2.py
MODE=train tools/run_prof.sh r06ev/r06_cfg2_train_kernel_stats python $R/tools/prof_cfg2.py
# 4. PMC: L2<->fabric traffic of the GEMM kernels (cfg 3 and cfg 5), SQ counters of the attention kernels
bash tools/pmc_step_traffic.sh r06ev/r06_gemm_traffic.json > /dev/null 2>&1
PMC_CONFIG=cfg5 bash tools/pmc_step_traffic.sh r06ev/r06_gemm_traffic_cfg5.json > /dev/null 2>&1
PMC_TAG=r06ev/r05 bash tools/pmc_attn2.sh 1 > /dev/null 2>&1
# 5. attention A/B (bitwise + timing) and the cfg-5 line on its own
timeout 300 python tools/ab_attn_kv3.py > $O/r06_ab_attn.txt 2>&1
timeout 300 python tools/time_attn.py > $O/r06_time_attn.txt 2>&1
timeout 600 python bench.py --config cfg5 --steps 6 --warmup 3 --no-cpu-baseline --no-forward-leg --no-cfg5-extra > $O/bench_r06_cfg5.json 2> /dev/null
ls -la $O
