# soak: thousands of consecutive steps per configuration (persistent kernels of round 4 + ride-along jobs in both launches): no time-out, finite loss
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 300 python -m pytest tests/test_gpu_edge.py -x -q > gpurun_out/r4/edge.log 2>&1; tail -2 gpurun_out/r4/edge.log
for a in "" "--ragged" "--arch 2" "--arch 2 --bf16" "--bf16"; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 800 --blocks 5 --warmup 10 $a > gpurun_out/r4/soak.json 2> gpurun_out/r4/soak.err; rc=$?
python3 -c "
import json
j=json.loads(open('gpurun_out/r4/soak.json').read().strip().splitlines()[-1])
print('soak $a rc=$rc', j['ms_per_step'], j['timed_blocks'], j.get('final_loss'))" || tail -3 gpurun_out/r4/soak.err
done
