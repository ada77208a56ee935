# SQ counters of the three operand layouts (NT forward, NN dgrad, TN wgrad) of the 8-wave (default: force 256, gemm_nt_256) or the
# 4-wave (bash tools/pmc_gemm.sh 4 gemm_w4) 256x256 GEMM kernel.  Run on the GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
FORCE=${1:-256}; KN=${2:-gemm_nt_256}
OUT=$R/gpurun_out/r03_pmc_gemm_layouts_force$FORCE.txt
: > $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  echo "== pass $i: $set" >> $OUT
  rm -rf /tmp/pg$i
  MH_GEMM_FORCE=$FORCE timeout 200 rocprofv3 --pmc $set -d /tmp/pg$i -o r -- python $R/tools/pmc_gemm.py > /tmp/pg$i.log 2>&1 || { echo "pass failed/timeout"; tail -3 /tmp/pg$i.log; } >> $OUT
  f=$(find /tmp/pg$i -name "*.db" 2>/dev/null | head -1)
  [ -n "$f" ] && python $R/tools/rocpd_pmc.py $f $KN >> $OUT 2>&1
done
