# usage: pbdbg2.sh "<bench flags>" "<env assignments>" dbg...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
flags="$1"; envs="$2"; shift; shift
for d in "$@"; do
  env $envs NVQA_PB_DBG=$d timeout -k 10 200 python bench.py $flags --no-secondary --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); k = r['kernel_ms_per_step']; print('$flags $envs NVQA_PB_DBG=$d', 'step', r['ms_per_step'], 'fwd', k['lstm_step_fwd'], 'bwd', k['lstm_step_bwd'])"
done 2>&1 | tee -a gpurun_out/r3/pbdbg.log
