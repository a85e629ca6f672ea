python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench3.log 2>&1; tail -c 4500 gpurun_out/r2_bench3.log
