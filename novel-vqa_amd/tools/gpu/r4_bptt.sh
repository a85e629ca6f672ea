cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 420 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_b500.py -x -q -k "persistent_bptt or default_batch_500 or headline or ride_along" > gpurun_out/r4/bptt.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r4/bptt.log
tail -15 gpurun_out/r4/bptt.log
[ $rc -eq 0 ] || exit $rc
NVQA_BWD_KERNEL=3 timeout -k 10 420 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_parity_r2.py -x -q -k "bf16 or ride_along" > gpurun_out/r4/bptt_bf16_v3.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r4/bptt_bf16_v3.log
tail -5 gpurun_out/r4/bptt_bf16_v3.log
[ $rc -eq 0 ] || exit $rc
for v in 3 2; do
NVQA_BWD_KERNEL=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r4/bench_v$v.json 2> gpurun_out/r4/bench_v$v.err
NVQA_BWD_KERNEL=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --arch 2 --bf16 > gpurun_out/r4/bench_bf16_v$v.json 2> gpurun_out/r4/bench_bf16_v$v.err
done
python - <<'PY'
import json
for f in ("bench_v3","bench_v2","bench_bf16_v3","bench_bf16_v2"):
    try:
        j = json.loads(open(f"gpurun_out/r4/{f}.json").read().strip().splitlines()[-1])
        print(f, j["ms_per_step"], j["timed_blocks"]["ms_per_step_min"], {k: v for k, v in j["kernel_ms_per_step"].items() if k.startswith("lstm")})
    except Exception as e:
        print(f, "failed", e)
PY
