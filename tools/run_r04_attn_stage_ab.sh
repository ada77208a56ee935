# Round 4, late: K|V / Q|dO tile copies through a buffer descriptor (attn_tiles.h stage_rows_buf) against the previous build of the attention
# kernels (tools/dev_arms/libmerlin_hip_attnprev.so, built by hand from the sources of commit d888e0c), then the attention tests.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_geometry_gpu.py tests/test_masks_gpu.py tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider -k "attn or attention or geometry or mask or ragged or golden or tiny or flash" > gpurun_out/attn_stage_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/attn_stage_tests.log
grep -n "passed\|failed" gpurun_out/attn_stage_tests.log | tail -3
for i in 1 2; do
  MH_LIB_PATH=$R/tools/dev_arms/libmerlin_hip_attnprev.so python tools/time_attn.py
  python tools/time_attn.py
done > gpurun_out/r04_attn_stage_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04_attn_stage_ab.txt
