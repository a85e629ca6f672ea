cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 420 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_b500.py tests/test_gpu_fullsize.py -x -q -k "persistent_bptt or default_batch_500 or headline or ride_along or full_size" > gpurun_out/r4/bptt.log 2>&1; rc=$?; tail -3 gpurun_out/r4/bptt.log
[ $rc -eq 0 ] || exit $rc
for m in "--ragged" ""; do
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary $m > gpurun_out/r4/rag.json 2> gpurun_out/r4/rag.err
python - "$m" <<'PY'
import json,sys
j = json.loads(open("gpurun_out/r4/rag.json").read().strip().splitlines()[-1])
print(sys.argv[1] or "full", j["ms_per_step"], {k: v for k, v in j["kernel_ms_per_step"].items() if k.startswith("lstm")})
PY
done
