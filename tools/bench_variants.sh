cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_bench_variants_final2.txt
echo "# variant | ms/step | tokens/s | peak HBM GB | GEMM TFLOP/s | loss" > $OUT
for v in "" "--config cfg2" "--recompute" "--config cfg3-ragged" "--fp8-train" "--config cfg5" "--config cfg5-bf16" "--fwd-only" "--fwd-only --fp8-forward"; do
  timeout 600 python bench.py $v --steps 3 --warmup 2 --no-cpu-baseline --no-forward-leg 2>/dev/null | tail -1 > /tmp/line.json
  python - "$v" <<'PY' >> $OUT
import json, sys
try:
    d = json.load(open("/tmp/line.json"))
    print(f"{sys.argv[1]:28s} {d['ms_per_step']} {d['value']} {d.get('peak_hbm_gb')} {d.get('roofline', {}).get('achieved')} {d['config'].get('loss')}")
except Exception as e:
    print(f"{sys.argv[1]:28s} FAILED {e}")
PY
done
cat $OUT
