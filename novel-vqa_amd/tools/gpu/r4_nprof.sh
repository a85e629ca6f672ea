cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
for n in 3 8 20; do
NVQA_BENCH_NPROF=$n timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r4/np.json 2> gpurun_out/r4/np.err
python - $n <<'PY'
import json,sys
j = json.loads(open("gpurun_out/r4/np.json").read().strip().splitlines()[-1])
print("nprof", sys.argv[1], j["ms_per_step"], j["roofline"]["frac"], {k: v for k, v in j["kernel_ms_per_step"].items() if k.startswith("lstm") or k=="gemm_wgrad"}, round(sum(j["kernel_ms_per_step"].values()),3))
PY
done
