cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_gpu_vgg.py -x -q > gpurun_out/r3/b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/b.log
tail -12 gpurun_out/r3/b.log


