cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
for e in "X=1" "NVQA_RIDE_FWD=0"; do
env $e timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --ragged > gpurun_out/r4/rag.json 2> gpurun_out/r4/rag.err
python - "$e" <<'PY'
import json,sys
j = json.loads(open("gpurun_out/r4/rag.json").read().strip().splitlines()[-1])
print(sys.argv[1], j["ms_per_step"], {k: v for k, v in j["kernel_ms_per_step"].items() if k.startswith("lstm") or k in ("gemm_head_fwd","head_prep","emb_fwd")})
PY
env $e NVQA_PF_DBG=32 timeout -k 10 120 python bench.py --steps 4 --warmup 1 --blocks 1 --no-cpu-baseline --no-secondary --no-roofline --ragged 2>&1 >/dev/null | grep "persistent forward"
done
