cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
bash novel-vqa_amd/tools/gpu/pbdbg.sh 0 11 2 1
timeout -k 10 600 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_dp_fullsize.py tests/test_gpu_dp_shim.py -x -q -m gpu -k "persistent_bptt or headline or dp_ or shim or collective" > gpurun_out/r3/t3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/t3.log
tail -15 gpurun_out/r3/t3.log
grep dp_fullsize gpurun_out/parity_r03.jsonl | tail -2
