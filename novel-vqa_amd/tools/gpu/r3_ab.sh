cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_arch2.py tests/test_gpu_bf16.py -x -q 2>&1 | tail -3
timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 50 --warmup 10 > /tmp/x.json && python3 -c "
import json
j=json.loads(open('/tmp/x.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], {k:(v['ms_per_step'], v['frac']) for k,v in j['phases'].items()})"
