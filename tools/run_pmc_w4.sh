cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out
OUT=/root/repo/gpurun_out/pmc_w4.txt
: > $OUT
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum"; do
  i=$((i+1))
  echo "== pass $i: $set" >> $OUT
  timeout 100 rocprofv3 --pmc $set -d /tmp/pmc$i -o r -- python /root/repo/tools/pmc_w4.py > /tmp/pmc$i.log 2>&1 || { echo "pass failed/timeout"; tail -3 /tmp/pmc$i.log; } >> $OUT
  f=$(find /tmp/pmc$i -name "*.db" 2>/dev/null | head -1)
  [ -n "$f" ] && python /root/repo/tools/rocpd_pmc.py $f gemm_nt >> $OUT 2>&1
  [ -n "$f" ] && python /root/repo/tools/rocpd_pmc.py $f Custom >> $OUT 2>&1
done
