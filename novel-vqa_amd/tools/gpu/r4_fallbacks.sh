# The -m gpu suite with the fast paths switched off, one at a time: the routes a failed persistent launch, an
# ineligible shape or an A/B switch falls back to must stay parity-green.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 600 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_bench_launch.py ${K:+-k "$K"} > gpurun_out/r4/fallback_$name.log 2>&1
  echo "$name: $(tail -1 gpurun_out/r4/fallback_$name.log)"
}
run fwd_ring NVQA_FWD_KERNEL=1 &&
run fwd3_ragged_by_ring NVQA_FWD3_RAGGED=0 &&
run fwd3_ragged_without_skips NVQA_FWD3_RAGGED=1 &&
run bwd_ring NVQA_BWD_KERNEL=2 &&
run bwd_direct_bf16 NVQA_BWD_KERNEL=3 &&
K="not timeout and not ride and not rides and not dp" &&
run persist_bwd_off NVQA_PERSIST_BWD=0 &&
run persist_off NVQA_PERSIST=0 &&
K="" &&
run riders_off NVQA_RIDE_GEMM=0 NVQA_TOK_IN_BPTT=0 NVQA_RIDE_FWD=0 NVQA_BIAS_IN_BPTT=0 NVQA_X0_B16=0
