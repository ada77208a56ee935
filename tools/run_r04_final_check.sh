# what the driver runs at round end, on the final tree: GPU suite, smoke, default bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r04_gputests_final.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_gputests_final.log
tail -4 gpurun_out/r04_gputests_final.log; grep FAILED gpurun_out/r04_gputests_final.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/bench_r04_final_noflags.json 2> /dev/null; python -c "
import json; d=json.load(open('gpurun_out/bench_r04_final_noflags.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
