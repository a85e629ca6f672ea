cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3/full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/full.log
tail -25 gpurun_out/r3/full.log
