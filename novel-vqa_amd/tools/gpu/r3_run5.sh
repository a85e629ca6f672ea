cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3/full2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/full2.log
tail -6 gpurun_out/r3/full2.log
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/r3/bench2.json 2> gpurun_out/r3/bench2.err; tail -c 3000 gpurun_out/r3/bench2.json
