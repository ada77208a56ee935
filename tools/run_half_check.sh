cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/half
for m in 0 1 0 1; do
  MH_W4_HALF=$m python bench.py --config cfg2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/half/cfg2_train_half$m.$RANDOM.json 2> gpurun_out/half/err.log
  MH_W4_HALF=$m python bench.py --config cfg2 --steps 30 --warmup 5 --no-cpu-baseline --fwd-only > gpurun_out/half/cfg2_fwd_half$m.$RANDOM.json 2>> gpurun_out/half/err.log
done
grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/half/cfg2_train_half0.*.json gpurun_out/half/cfg2_train_half1.*.json gpurun_out/half/cfg2_fwd_half0.*.json gpurun_out/half/cfg2_fwd_half1.*.json
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/half/pytest_gpu.txt
cat gpurun_out/half/pytest_gpu.txt
