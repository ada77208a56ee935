R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_w4_gpu.py tests/test_ops_gpu.py -q -p no:cacheprovider -k "w4 or rope or swiglu or gemm" > gpurun_out/r04_w4_tests2.log 2>&1; echo "rc=$?" >> gpurun_out/r04_w4_tests2.log
tail -6 gpurun_out/r04_w4_tests2.log
python tools/ab_w4_forms.py > gpurun_out/r04_w4_forms_ab2.txt 2>&1; cat gpurun_out/r04_w4_forms_ab2.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
