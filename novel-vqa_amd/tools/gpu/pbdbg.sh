# ablation of the persistent BPTT kernel: NVQA_PB_DBG bits 1 no counter waits, 2 no cell math / stores, 8 A loads without memory traffic
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for d in "$@"; do
  NVQA_PB_DBG=$d timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); k = r['kernel_ms_per_step']; print('NVQA_PB_DBG=$d', 'step', r['ms_per_step'], 'fwd', k['lstm_step_fwd'], 'bwd', k['lstm_step_bwd'])" 
done 2>&1 | tee -a gpurun_out/r3/pbdbg.log
