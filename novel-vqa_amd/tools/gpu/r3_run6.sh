cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
NVQA_PB_V=2 timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py -q -m gpu > gpurun_out/r3/t_bf16.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/t_bf16.log
tail -5 gpurun_out/r3/t_bf16.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity_r2.py -q -m gpu -k "persistent or headline" > gpurun_out/r3/t6.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/t6.log
tail -3 gpurun_out/r3/t6.log
bash novel-vqa_amd/tools/gpu/pbdbg.sh 0
