python -m pytest tests/test_fp8_training_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4
for m in 3 11 3 11; do echo "== cfg5 MH_W4_MASK=$m"; MH_W4_MASK=$m python bench.py --config cfg5 --steps 5 --warmup 3 --no-cpu-baseline --no-forward-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['gemm_share_of_step'], d['config']['loss'])"; done
for m in 3 11; do echo "== cfg3 --fp8-train MH_W4_MASK=$m"; MH_W4_MASK=$m python bench.py --fp8-train --steps 5 --warmup 3 --no-cpu-baseline --no-forward-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['gemm_share_of_step'], d['config']['loss'])"; done
