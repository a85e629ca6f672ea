cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
for d in 128 130 136 129; do
NVQA_PF_DBG=$d timeout -k 10 120 python bench.py --steps 10 --warmup 3 --blocks 2 --no-cpu-baseline --no-secondary > gpurun_out/r4/abl.json 2> gpurun_out/r4/abl.err
python - $d <<'PY'
import json,sys
try:
    j = json.loads(open("gpurun_out/r4/abl.json").read().strip().splitlines()[-1])
    print("PF_DBG", sys.argv[1], j["ms_per_step"], {k: v for k, v in j["kernel_ms_per_step"].items() if k.startswith("lstm")})
except Exception as e:
    print(sys.argv[1], "failed", e, open("gpurun_out/r4/abl.err").read()[-300:])
PY
grep "nvqa\]" gpurun_out/r4/abl.err | head -5
done
