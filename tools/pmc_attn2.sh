# SQ counters of the product attention kernels (forward, dK/dV, dQ) at the cfg-3 geometry.  Run on the GPU box: bash tools/pmc_attn2.sh [causal 1|0]   (MH_ATTN_FWD_FORM=0|1|2 selects the forward form, PMC_TAG the output prefix)
R=${GRAFT_REPO_ROOT:-/root/repo}; C=${1:-1}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/${PMC_TAG:-r04}_pmc_attn_causal${C}_form${MH_ATTN_FWD_FORM:-0}.txt
: > $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  echo "== causal=$C pass $i: $set" >> $OUT
  rm -rf /tmp/pa$i
  timeout 200 rocprofv3 --pmc $set -d /tmp/pa$i -o r -- python $R/tools/prof_attn2.py $C > /tmp/pa$i.log 2>&1 || { echo "pass failed/timeout"; tail -3 /tmp/pa$i.log; } >> $OUT
  f=$(find /tmp/pa$i -name "*.db" 2>/dev/null | head -1)
  [ -n "$f" ] && python $R/tools/rocpd_pmc.py $f attn >> $OUT 2>&1
done
cat $OUT
