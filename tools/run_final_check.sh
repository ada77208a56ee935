cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r05ev
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/r05ev/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> gpurun_out/r05ev/pytest_gpu.txt
bash tools/run_r05_evidence.sh > gpurun_out/r05ev/bundle.log 2>&1
cat gpurun_out/r05ev/pytest_gpu.txt; tail -c 600 gpurun_out/r05ev/bench_r05_default.json
