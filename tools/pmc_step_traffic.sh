#!/bin/bash
# Regenerates profiles/$OUT: L2<->fabric traffic of every GEMM launch of one cfg-3 training step, from
# separate rocprofv3 --pmc passes (no tracing combined with --pmc) over `bench.py --steps 1 --warmup 1`.
# Run on the GPU box from the repo root: bash tools/pmc_step_traffic.sh   (writes gpurun_out/$OUT)
# PMC_CONFIG=cfg5 bash tools/pmc_step_traffic.sh r05_gemm_traffic_cfg5.json: the same for the fp8 step of cfg 5 (bench.py quotes it in extras.cfg5).
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=${1:-r06_gemm_traffic.json}; cd /tmp && export TMPDIR=/tmp
CFGARGS=""; [ "${PMC_CONFIG:-cfg3}" = "cfg5" ] && CFGARGS="--config cfg5"
mkdir -p $R/gpurun_out
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmcs$i
  timeout 600 rocprofv3 --pmc $set -d /tmp/pmcs$i -o r -- python $R/bench.py $CFGARGS --steps 1 --warmup 1 --no-cpu-baseline --no-forward-leg --no-extras --no-cfg5-extra > /tmp/pmcs$i.log 2>&1 || echo "pass $i failed (counter set: $set)"
done
# (the summary is stamped with a hash of the kernel sources; bench.py quotes it only while that hash matches its own build)
python $R/tools/pmc_step_traffic.py /tmp/pmcs1 /tmp/pmcs2 /tmp/pmcs3 > $R/gpurun_out/$OUT
cat $R/gpurun_out/$OUT
