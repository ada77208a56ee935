R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r04_gputests_6.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_gputests_6.log
tail -4 gpurun_out/r04_gputests_6.log
bash tools/run_r04_evidence.sh v2
