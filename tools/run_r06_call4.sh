#!/bin/bash
# round 6, GPU call 4: N > 1 readiness on one GPU - (a) two RCCL ranks on one device, (b) a one-rank RCCL group's all_reduce, (c) bench.py --force-dp (the N > 1 code path with an
# RCCL group of one rank: GradSync on its stream, per-rank gathers, RCCL tuning-log parse)
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r06c4; mkdir -p $O
timeout 300 python tools/probes/rccl_two_ranks_one_gpu.py > $O/rccl_probe.txt 2>&1; cat $O/rccl_probe.txt | grep -v amdgpu.ids | tail -12
timeout 600 python bench.py --force-dp --steps 6 --warmup 3 --no-cpu-baseline --no-extras --no-cfg5-extra --no-forward-leg > $O/bench_force_dp.json 2> $O/bench_force_dp.err; echo "force-dp rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06c4/bench_force_dp.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("ms_per_step", "rccl_ranks", "comm_ms_total", "comm_ms_exposed", "comm_collectives_per_step", "comm_gb_per_step", "rccl_choice", "hbm_headroom_gb", "mem_level", "recompute_fallback")})
PY
