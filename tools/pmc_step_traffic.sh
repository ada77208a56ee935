#!/bin/bash
# Regenerates profiles/$OUT: L2<->fabric traffic of every GEMM launch of one cfg-3 training step, from
# separate rocprofv3 --pmc passes (no tracing combined with --pmc) over `bench.py --steps 1 --warmup 1`.
# Run on the GPU box from the repo root: bash tools/pmc_step_traffic.sh   (writes gpurun_out/$OUT)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=${1:-r04_gemm_traffic.json}; cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmcs$i
  timeout 600 rocprofv3 --pmc $set -d /tmp/pmcs$i -o r -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-forward-leg --no-extras > /tmp/pmcs$i.log 2>&1 || echo "pass $i failed (counter set: $set)"
done
# (the summary is stamped with a hash of the kernel sources; bench.py quotes it only while that hash matches its own build)
python $R/tools/pmc_step_traffic.py /tmp/pmcs1 /tmp/pmcs2 /tmp/pmcs3 > $R/gpurun_out/$OUT
cat $R/gpurun_out/$OUT
