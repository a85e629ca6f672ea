cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
bash novel-vqa_amd/tools/gpu/pbdbg.sh 0 11 2 1
timeout -k 10 420 python -m pytest tests/test_gpu_parity_r2.py -x -q -m gpu -k "persistent_bptt or headline" > gpurun_out/r3/t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/t2.log
tail -5 gpurun_out/r3/t2.log
