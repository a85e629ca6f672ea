cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/r4/full.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r4/full.log
tail -30 gpurun_out/r4/full.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench_sec.json 2> gpurun_out/r4/bench_sec.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r4/bench_sec.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["value"], j["timed_blocks"], j["persistent"])
for k, v in j.get("secondary", {}).items():
    print(k, {a: b for a, b in v.items() if a in ("value", "ms_per_step", "ms_per_step_min_max", "ms_per_batch", "mfma_frac")})
print(j.get("cpu_baseline"))
PY
