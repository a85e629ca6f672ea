cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3/full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/full.log
tail -8 gpurun_out/r3/full.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 | tail -c 900
timeout -k 10 300 python bench.py --arch 2 --bf16 --no-cpu-baseline --no-secondary --steps 20 --warmup 5 | tail -c 700
