cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
for ks in 0 2 4 8 16; do
NVQA_WB_KS=$ks timeout -k 10 120 python bench.py --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline --no-secondary --arch 2 --bf16 > gpurun_out/r4/ks.json 2> gpurun_out/r4/ks.err
python - $ks <<'PY'
import json,sys
j = json.loads(open("gpurun_out/r4/ks.json").read().strip().splitlines()[-1])
print("NVQA_WB_KS", sys.argv[1], j["ms_per_step"], {k: v for k, v in j["kernel_ms_per_step"].items() if k in ("gemm_wgrad","reduce_slabs")})
PY
done
