for dbg in 0 16 1 3 11; do
NVQA_PERSIST=1 NVQA_PF_DBG=$dbg python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('dbg',$dbg, d['ms_per_step'], d['kernel_ms_per_step']['lstm_step_fwd'])"
done
NVQA_PERSIST=1 python -m pytest tests/test_gpu_parity_r2.py -m gpu -q -x -k persistent 2>&1 | tail -3
