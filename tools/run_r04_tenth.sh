R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_w4_gpu.py tests/test_fp8_training_gpu.py -q -x -p no:cacheprovider > gpurun_out/r04_w4_tests3.log 2>&1; echo "rc=$?" >> gpurun_out/r04_w4_tests3.log
tail -12 gpurun_out/r04_w4_tests3.log
python tools/ab_w4_forms.py 2>&1 | grep -v amdgpu > gpurun_out/r04_w4_forms_ab3.txt; cat gpurun_out/r04_w4_forms_ab3.txt
for M in default 251; do
  if [ $M = default ]; then unset MH_W4_MASK; else export MH_W4_MASK=$M; fi
  python bench.py --config cfg5 --steps 6 --warmup 3 --no-cpu-baseline --no-forward-leg > gpurun_out/bench_r04_cfg5_mask_$M.json 2>> gpurun_out/bench_r04_cfg5_ab.err
  python -c "
import json; d=json.load(open('gpurun_out/bench_r04_cfg5_mask_$M.json')); print('cfg5 mask=$M', d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['config']['loss'])"
done
unset MH_W4_MASK
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/bench_r04_rcp.json 2> /dev/null
python -c "
import json; d=json.load(open('gpurun_out/bench_r04_rcp.json')); print('cfg3', d['ms_per_step'], d['roofline']['achieved'], d['forward_only']['ms_per_step'], d['config']['loss'])"
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r04_gputests_7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_gputests_7.log
tail -4 gpurun_out/r04_gputests_7.log; grep FAILED gpurun_out/r04_gputests_7.log | head
