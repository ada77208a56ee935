# Round 4, late: register-resident row kernels (fp32-stream reader, quantising RMSNorm, row quantiser).  Tests of the touched ops and of the
# model on top of them, then the whole evidence bundle (tools/run_r04_evidence.sh: gemm.hip holds the row quantiser, so the traffic stamp moves).
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-v6}
cd $R; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_fp8_training_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_parity_floor_gpu.py -x -q -m gpu -p no:cacheprovider > gpurun_out/residue_tests_$TAG.log 2>&1
echo "pytest rc=$?" >> gpurun_out/residue_tests_$TAG.log
tail -4 gpurun_out/residue_tests_$TAG.log
bash tools/run_r04_evidence.sh $TAG
grep -h "norm_fwd_f32in_k\|rmsnorm_fwd_q8_k\|quant_fp8_rows_k" gpurun_out/r04_step_cfg3_kernel_stats_$TAG.txt gpurun_out/r04_step_cfg5_fp8_kernel_stats_$TAG.txt | cut -c1-50,110-190
