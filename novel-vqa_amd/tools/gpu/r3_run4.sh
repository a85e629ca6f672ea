cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_wgrad_bf16.py tests/test_gpu_end_to_end.py tests/test_gpu_vgg.py tests/test_gpu_parity_r2.py tests/test_gpu_dp_fullsize.py -q -m gpu -k "short_cascade or wgrad_bf16 or full_width or preprocess or fc7_matches or timeout or collective" > gpurun_out/r3/t4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/t4.log
tail -30 gpurun_out/r3/t4.log
grep -E "bf16_short|wgrad_bf16|e2e_full|dp_fullsize" gpurun_out/parity_r03.jsonl | tail -8
