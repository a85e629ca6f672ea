# soak: thousands of consecutive steps per configuration (persistent kernels + ride-along jobs): no time-out, finite loss
cd $GRAFT_REPO_ROOT
for a in "" "--ragged" "--arch 2" "--arch 2 --bf16" "--bf16"; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 4000 --warmup 10 $a > /tmp/x.json 2> /tmp/x.err; rc=$?
python3 -c "
import json
j=json.loads(open('/tmp/x.json').read().strip().splitlines()[-1])
print('soak $a rc=$rc', j['ms_per_step'], j.get('final_loss'))" || tail -3 /tmp/x.err
done
