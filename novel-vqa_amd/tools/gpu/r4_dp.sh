cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_dp.py tests/test_gpu_dp_shim.py tests/test_gpu_dp_fullsize.py tests/test_gpu_bench_launch.py -x -q > gpurun_out/r4/dp.log 2>&1; rc=$?; tail -5 gpurun_out/r4/dp.log
grep dp_fullsize gpurun_out/parity_r03.jsonl | tail -4
python __graft_entry__.py smoke 2>&1 | tail -2
exit $rc
