#!/bin/bash
# sanity of the bench variants the round-end driver does not run itself (one line per variant: ms per step, loss)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/variants
run() { name=$1; shift; timeout 600 python bench.py "$@" --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-cfg5-extra --no-forward-leg > gpurun_out/variants/$name.json 2> gpurun_out/variants/$name.err; python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/variants/{n}.json").read().strip().splitlines()[-1])
    print(f"{n:14s} {d['ms_per_step']:9.2f} ms  loss {d.get('loss')}  comm {d.get('comm_ms_exposed')}  mem_level {d.get('mem_level')}")
except Exception as e:
    print(n, "FAILED", e, open(f"gpurun_out/variants/{n}.err").read()[-400:])
PY
}
run default
run force_dp --force-dp
run ragged --config cfg3-ragged
run fp8_cfg3 --fp8-train
run recompute --recompute
run mem2 --mem-level 2
run cfg2 --config cfg2
