cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for v in 1 2; do
NVQA_PB_V=$v timeout -k 10 300 python bench.py --arch 2 --bf16 --no-secondary --no-cpu-baseline --steps 20 --warmup 5 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); k = r['kernel_ms_per_step']; print('arch2 bf16 PB_V=$v', 'step', r['ms_per_step'], {a: k[a] for a in k if k[a] > 0.03})"
done
NVQA_PB_V=2 timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu > gpurun_out/r3/t_bf16.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3/t_bf16.log
tail -5 gpurun_out/r3/t_bf16.log
