cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3/full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/full.log
tail -6 gpurun_out/r3/full.log
timeout -k 10 500 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r3/bench_sec.json 2> gpurun_out/r3/bench_sec.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r3/bench_sec.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["value"])
for k, v in j.get("secondary", {}).items():
    print(k, {a: b for a, b in v.items() if a in ("value", "ms_per_step", "ms_per_batch", "mfma_frac")})
PY
