#!/bin/bash
# round 4, closing: the direct-operand forward kernel on B = 500 and (NVQA_FWD3_RAGGED=1: its instance without skips) on ragged batches
set -o pipefail
O=gpurun_out/r4_fwd3_ragged; rm -rf $O; mkdir -p $O
T="tests/test_gpu_b500.py tests/test_gpu_arch1.py tests/test_gpu_parity_r2.py tests/test_gpu_edge.py"
timeout -k 10 500 python -m pytest $T -m gpu -x -q > $O/tests_default.log 2>&1; echo "default rc $?" | tee -a $O/summary.txt; tail -3 $O/tests_default.log
timeout -k 10 300 python tests/dbg_rag_rows.py > $O/dbg_rag_rows.log 2>&1; cat $O/dbg_rag_rows.log
for m in 2; do
  NVQA_FWD3_RAGGED=$m timeout -k 10 500 python -m pytest $T -m gpu -x -q > $O/tests_ragged$m.log 2>&1; echo "ragged=$m rc $?" | tee -a $O/summary.txt; tail -3 $O/tests_ragged$m.log
done
for m in 0 2; do
  NVQA_FWD3_RAGGED=$m timeout -k 10 200 python bench.py --ragged --no-cpu-baseline --no-secondary --steps 50 --warmup 10 > $O/bench_ragged_mode$m.json 2> $O/bench_ragged_mode$m.err
  echo "ragged bench mode $m rc $?" | tee -a $O/summary.txt
  python -c "import json,sys; d=json.loads(open('$O/bench_ragged_mode$m.json').read().strip().splitlines()[-1]); print('mode $m ms_per_step', d['ms_per_step'], d.get('ms_per_step_blocks'), d.get('persistent'))" | tee -a $O/summary.txt
done
