# SQ counters of the three operand layouts of gemm_nt_256 (NT forward, NN dgrad, TN wgrad): where the K-strided forms lose time.  Run on the GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/r02_pmc_gemm_layouts.txt
: > $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  echo "== pass $i: $set" >> $OUT
  rm -rf /tmp/pg$i
  timeout 200 rocprofv3 --pmc $set -d /tmp/pg$i -o r -- python $R/tools/pmc_gemm.py > /tmp/pg$i.log 2>&1 || { echo "pass failed/timeout"; tail -3 /tmp/pg$i.log; } >> $OUT
  f=$(find /tmp/pg$i -name "*.db" 2>/dev/null | head -1)
  [ -n "$f" ] && python $R/tools/rocpd_pmc.py $f gemm_nt_256 >> $OUT 2>&1
done
