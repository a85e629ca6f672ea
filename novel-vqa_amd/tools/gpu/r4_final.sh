cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
rm -f gpurun_out/parity_r04.jsonl
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=5 > gpurun_out/r4/full.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r4/full.log
tail -9 gpurun_out/r4/full.log
[ $rc -eq 0 ] || exit $rc
python __graft_entry__.py smoke 2>&1 | tail -1
timeout -k 10 600 python bench.py > gpurun_out/r4/bench_default.json 2> gpurun_out/r4/bench_default.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r4/bench_default.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["value"], j["timed_blocks"], j["persistent"], j["roofline"]["frac"], j["step_mfma_frac"])
for k, v in j.get("secondary", {}).items():
    print(k, {a: b for a, b in v.items() if a in ("value", "ms_per_step", "ms_per_step_min_max", "ms_per_batch", "mfma_frac")})
print(j.get("cpu_baseline"))
print(j["kernel_ms_per_step"])
PY
