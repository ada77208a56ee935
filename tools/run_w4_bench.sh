python -m pytest tests/test_masks_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
for m in 0 1 3 7 0 3; do echo "== MH_W4_MASK=$m"; MH_W4_MASK=$m python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras --no-forward-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['gemm_share_of_step'])"; done
