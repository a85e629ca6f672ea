# the GPU suite with the persistent kernels switched off (per-level fallback paths), and with the ride-along jobs in their own launches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
NVQA_PERSIST_BWD=0 timeout -k 10 1000 python -m pytest tests -x -q -m gpu -k "not timeout and not ride and not dp" > gpurun_out/r3/fb1.log 2>&1; echo "rc=$?" >> gpurun_out/r3/fb1.log; tail -3 gpurun_out/r3/fb1.log
NVQA_PERSIST=0 timeout -k 10 1000 python -m pytest tests -x -q -m gpu -k "not timeout and not ride and not dp" > gpurun_out/r3/fb2.log 2>&1; echo "rc=$?" >> gpurun_out/r3/fb2.log; tail -3 gpurun_out/r3/fb2.log
NVQA_RIDE_GEMM=0 NVQA_TOK_IN_BPTT=0 timeout -k 10 1000 python -m pytest tests -x -q -m gpu -k "not dp" > gpurun_out/r3/fb3.log 2>&1; echo "rc=$?" >> gpurun_out/r3/fb3.log; tail -3 gpurun_out/r3/fb3.log
