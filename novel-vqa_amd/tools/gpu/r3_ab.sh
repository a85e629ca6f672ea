cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_bf16.py tests/test_gpu_arch2.py -x -q 2>&1 | tail -3
for r in 1 2; do
for a in "" "--arch 2 --bf16" "--ragged"; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 50 --warmup 10 $a > /tmp/x.json && python3 -c "
import json
j=json.loads(open('/tmp/x.json').read().strip().splitlines()[-1])
print('$a', j['ms_per_step'], j['phases']['lstm_step_bwd']['ms_per_step'])"
done
done
