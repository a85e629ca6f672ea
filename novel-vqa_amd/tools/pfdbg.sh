timeout -k 5 300 python -m pytest tests/test_gpu_parity_r2.py -m gpu -q -x -k "headline or persistent or satbias" 2>&1 | tail -3
for dbg in 0 1 3; do
NVQA_PB_DBG=$dbg timeout -k 5 100 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('dbg',$dbg, d['ms_per_step'], d['kernel_ms_per_step']['lstm_step_bwd'])"
done
