# SQ counters of the attention kernels: where the wave cycles go (issuing / parked / issue-stalled, MFMA busy, LDS).  Run on the GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/r02_pmc_attn_arms.txt
: > $OUT
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/sq_counters.txt
for w in 0; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM"; do
    i=$((i+1))
    echo "== arm wide=$w pass $i: $set" >> $OUT
    rm -rf /tmp/pm$w$i
    timeout 200 rocprofv3 --pmc $set -d /tmp/pm$w$i -o r -- python $R/tools/prof_attn.py $w > /tmp/pm$w$i.log 2>&1 || { echo "pass failed/timeout"; tail -3 /tmp/pm$w$i.log; } >> $OUT
    f=$(find /tmp/pm$w$i -name "*.db" 2>/dev/null | head -1)
    [ -n "$f" ] && python $R/tools/rocpd_pmc.py $f attn >> $OUT 2>&1
  done
done
