# same-box A/B of the final build: default | round-3 dispatch (8-wave fused forms, per-problem CLIP wgrads) | 16-bit residual streams
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for v in final r3dispatch streams16 final2; do
  case $v in
    final|final2) env -u MH_W4_MASK python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/bench_r04_ab_$v.json 2>> gpurun_out/bench_r04_ab.err;;
    r3dispatch) MH_W4_MASK=11 MH_WGRAD_GROUPED=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/bench_r04_ab_$v.json 2>> gpurun_out/bench_r04_ab.err;;
    streams16) python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --residual-16bit > gpurun_out/bench_r04_ab_$v.json 2>> gpurun_out/bench_r04_ab.err;;
  esac
  python -c "
import json; d=json.load(open('gpurun_out/bench_r04_ab_$v.json')); print('$v', d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['forward_only']['ms_per_step'], d['config']['loss'])"
done
