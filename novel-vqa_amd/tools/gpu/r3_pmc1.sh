cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3/pmc1
mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
grep -i -E "icache|ifetch|inst_cache|SQC_" $O/counters.txt | head -40
B="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES --output-format csv -d $O/sq -- $B > /dev/null 2> $O/sq.log
cd $R
python3 - <<'PY'
import csv,glob,collections,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r3/pmc1'
f=(glob.glob(O+'/sq/*/*counter_collection.csv')+glob.glob(O+'/sq/*counter_collection.csv'))[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'][:60]
    if 'persist' not in k: continue
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVE_CYCLES': n[k]+=1
for k in agg:
    a={c:v/n[k] for c,v in agg[k].items()}
    wc=a['SQ_WAVE_CYCLES']
    print(k, n[k], {c:round(v) for c,v in a.items()})
    print('   parked %.3f issue_stall %.3f issuing %.3f lds_conf %.4f'%(a['SQ_WAIT_ANY']/wc,a['SQ_WAIT_INST_ANY']/wc,a['SQ_ACTIVE_INST_ANY']/wc,a['SQ_LDS_BANK_CONFLICT']/max(a['SQ_LDS_IDX_ACTIVE'],1)))
PY
