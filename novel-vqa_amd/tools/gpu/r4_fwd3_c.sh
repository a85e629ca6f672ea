cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_parity_r2.py -x -q -m gpu -k "persistent or headline or timeout or ride" > gpurun_out/r4/fwd3_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r4/fwd3_tests.log
[ $rc -eq 0 ] || exit $rc
for d in 0 128; do
NVQA_PF_DBG=$d timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r4/fwd3_bench_3.json 2> gpurun_out/r4/fwd3_bench_3.err || { tail -5 gpurun_out/r4/fwd3_bench_3.err; exit 1; }
python - <<PY
import json
j = json.loads(open("gpurun_out/r4/fwd3_bench_3.json").read().strip().splitlines()[-1])
print("PF_DBG=$d", j["ms_per_step"], j["timed_blocks"], j["roofline"]["frac"], {a: b for a, b in j["kernel_ms_per_step"].items() if "lstm" in a})
PY
grep "nvqa\]" gpurun_out/r4/fwd3_bench_3.err | head -4
done
