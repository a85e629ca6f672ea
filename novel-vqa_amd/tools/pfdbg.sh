#!/bin/bash
# ablation of the persistent kernels (NVQA_PF_DBG / NVQA_PB_DBG bits: 1 no flag waits, 2 no cell, 8 loads without traffic)
# usage: pfdbg.sh "<bench flags>" PF|PB dbg...
flags="$1"; var="NVQA_$2_DBG"; shift; shift
for d in "$@"; do
  env $var=$d python bench.py $flags --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); k = r['kernel_ms_per_step']; print('$var=$d', 'step', r['ms_per_step'], 'fwd', k['lstm_step_fwd'], 'bwd', k['lstm_step_bwd'])"
done
