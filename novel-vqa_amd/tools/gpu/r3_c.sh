cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_bf16.py tests/test_gpu_dp_fullsize.py tests/test_gpu_dp.py tests/test_gpu_variants.py -x -q > gpurun_out/r3/c.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/c.log
tail -4 gpurun_out/r3/c.log
show() { python3 -c "
import json,sys
j=json.loads(open('$1').read().strip().splitlines()[-1])
print('$2', j['ms_per_step'])
"; }
for a in "" "--arch 2" "--arch 2 --bf16"; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 50 --warmup 10 $a > /tmp/x.json && show /tmp/x.json "side $a"
NVQA_TOK_SIDE=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 50 --warmup 10 $a > /tmp/x.json && show /tmp/x.json "main $a"
done
