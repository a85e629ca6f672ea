# round-3 closing run: full GPU suite, the driver-shaped bench line, the profile sets
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/final/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final/pytest.log
tail -4 gpurun_out/final/pytest.log
cp gpurun_out/parity_r03.jsonl gpurun_out/final/parity_r03.jsonl
timeout -k 10 600 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --arch 2 --bf16 --no-secondary > gpurun_out/final/bench_bf16_arch2.json 2>> gpurun_out/final/bench_default.err
timeout -k 10 400 bash novel-vqa_amd/tools/profile_r03.sh > gpurun_out/final/prof_r03.log 2>&1 && echo headline ok &&
timeout -k 10 400 bash novel-vqa_amd/tools/profile_r03.sh bf16_arch2 "--arch 2 --bf16" > gpurun_out/final/prof_r03_bf16.log 2>&1 && echo bf16 ok &&
timeout -k 10 400 bash novel-vqa_amd/tools/profile_vgg_r03.sh f32 64 > gpurun_out/final/prof_vgg_f32.log 2>&1 && echo vgg ok &&
timeout -k 10 400 bash novel-vqa_amd/tools/profile_vgg_r03.sh bf16 64 bf16 > gpurun_out/final/prof_vgg_bf16.log 2>&1 && echo vggbf16 ok
python3 - <<'PY'
import json
j = json.loads(open("gpurun_out/final/bench_default.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["value"], j["roofline"]["frac"], j.get("cpu_baseline"))
for k, v in j.get("secondary", {}).items():
    print(k, {a: b for a, b in v.items() if a in ("value", "ms_per_step", "ms_per_batch", "mfma_frac")})
PY
