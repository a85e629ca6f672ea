cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout -k 10 600 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err; echo "bench rc=$?"
python3 - <<'PY'
import json
j = json.loads(open("gpurun_out/final/bench_default.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["value"], j["roofline"]["frac"])
for k, v in j.get("secondary", {}).items():
    print(k, {a: b for a, b in v.items() if a in ("value", "ms_per_step", "ms_per_batch", "mfma_frac")})
PY
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()"
