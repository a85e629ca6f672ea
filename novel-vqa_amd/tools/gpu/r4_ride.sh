cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 300 python -m pytest tests/test_gpu_golden.py -x -q > gpurun_out/r4/golden.log 2>&1; tail -2 gpurun_out/r4/golden.log
NVQA_PERSIST_BWD=0 timeout -k 10 300 python -m pytest tests/test_gpu_golden.py -x -q > gpurun_out/r4/golden2.log 2>&1; tail -2 gpurun_out/r4/golden2.log
for e in "X=1" "NVQA_RIDE_GEMM=0"; do
echo "== $e"
env $e NVQA_PB_DBG=32 timeout -k 10 120 python bench.py --steps 4 --warmup 1 --blocks 1 --no-cpu-baseline --no-secondary --no-roofline 2>&1 >/dev/null | grep "persistent BPTT"
env $e timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r4/ride.json 2> gpurun_out/r4/ride.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r4/ride.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["timed_blocks"], {k: v for k, v in j["kernel_ms_per_step"].items() if k in ("lstm_step_bwd","gemm_head_bwd","colsum")}, j.get("ridden_head_gflop_per_step"))
PY
done
