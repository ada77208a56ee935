#!/bin/bash
# round 6, GPU call 2: the headline-size test (fixed engine configuration), the cfg-2 raster-group A/B, the vendor yardstick re-run on this tree
# and the FULL names of the kernels the vendor library picks for the four decoder NT shapes (comparator only: tools/yardstick.py, tools/yard_names.py)
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r06c2; mkdir -p $O
timeout 1500 python -m pytest tests/test_headline_gpu.py -x -q -s -m gpu > $O/headline.log 2>&1; echo "headline rc=$?" | tee -a $O/headline.log
tail -8 $O/headline.log
timeout 600 python tools/ab_cfg2_raster.py 4 5 8 3 > $O/r06_cfg2_raster_ab.txt 2> $O/raster.err; cat $O/r06_cfg2_raster_ab.txt
timeout 600 python tools/yardstick.py > $O/r06_yardstick.txt 2> $O/yard.err; cat $O/r06_yardstick.txt
R=$PWD; P=$R/$O/prof_names; mkdir -p $P
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $P -- python $R/tools/yard_names.py > $P/log.txt 2>&1 )
db=$(find $P -name "*_results.db" | head -1)
python - "$db" > $O/r06_vendor_kernel_names.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for n, k, a in c.execute("select name, count(*), avg(duration) from kernels group by name order by sum(duration) desc limit 12"):
    print(f"{k:4d} calls  avg {a/1e6:8.4f} ms  {n}")
PY
cat $O/r06_vendor_kernel_names.txt | cut -c1-400
rm -rf $P
