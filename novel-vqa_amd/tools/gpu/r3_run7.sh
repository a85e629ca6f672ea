cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_bf16.py tests/test_gpu_arch2.py -x -q -m gpu -k "persistent or headline or bf16 or arch2 or quirk" > gpurun_out/r3/t7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/t7.log
tail -4 gpurun_out/r3/t7.log
bash novel-vqa_amd/tools/gpu/pbdbg2.sh "" "NVQA_X=0" 0 2 11
bash novel-vqa_amd/tools/gpu/pbdbg2.sh "--arch 2 --bf16" "NVQA_X=0" 0
bash novel-vqa_amd/tools/gpu/pbdbg2.sh "--ragged" "NVQA_X=0" 0
bash novel-vqa_amd/tools/gpu/pbdbg2.sh "--arch 2" "NVQA_X=0" 0
