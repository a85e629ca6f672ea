set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 420 python -m pytest tests/test_gpu_parity_r2.py -x -q -m gpu -k "persistent_bptt or headline or persistent_forward" > gpurun_out/r3/t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/t1.log
tail -5 gpurun_out/r3/t1.log
timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r3/b1_new.log 2>&1 ; tail -1 gpurun_out/r3/b1_new.log | cut -c1-1500
NVQA_PB_V=1 timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r3/b1_old.log 2>&1 ; tail -1 gpurun_out/r3/b1_old.log | cut -c1-600
NVQA_PERSIST_BWD=0 timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r3/b1_lvl.log 2>&1 ; tail -1 gpurun_out/r3/b1_lvl.log | cut -c1-600
NVQA_PB_DBG=32 timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r3/b1_ts.log 2>&1 ; grep nvqa gpurun_out/r3/b1_ts.log
