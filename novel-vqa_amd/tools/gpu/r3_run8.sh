cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_gpu_dp.py tests/test_gpu_dp_shim.py tests/test_gpu_dp_fullsize.py tests/test_gpu_wgrad_bf16.py -q -m gpu > gpurun_out/r3/t8.log 2>&1; echo "rc=$?" >> gpurun_out/r3/t8.log
tail -5 gpurun_out/r3/t8.log
