#!/bin/bash
# usage: tools/run_prof.sh <out-name> <command...>   (on the GPU box, from the repo root): rocprofv3 --kernel-trace --stats of the command,
# summary -> gpurun_out/<out-name>.txt (tools/rocpd_stats.py on the results database)
set -u
name=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_$name
mkdir -p "$out"
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$out" -- "$@" > "$out/log.txt" 2>&1 )
db=$(find "$out" -name "*_results.db" | head -1)
if [ -n "$db" ]; then python "$root/tools/rocpd_stats.py" "$db" > "$root/gpurun_out/$name.txt" 2>&1; rm -rf "$out"; else echo "no results db (see $out/log.txt)"; fi
