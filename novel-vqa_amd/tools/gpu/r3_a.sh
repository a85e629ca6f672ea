cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_bf16.py -x -q > gpurun_out/r3/a.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/a.log
tail -3 gpurun_out/r3/a.log
show() { python3 -c "
import json,sys
j=json.loads(open('$1').read().strip().splitlines()[-1])
print('$1', j['ms_per_step'], {k:v['ms_per_step'] for k,v in j['phases'].items()}, {k:v['ms_per_step'] for k,v in j['hbm_kernels'].items()})
"; }
for a in "" "--arch 2 --bf16"; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 5 $a > /tmp/base.json && show /tmp/base.json
done
