# usage: r3_pmc2.sh <tag> [env assignments...]  -- SQ + instruction-cache counters of the persistent kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
for e in "$@"; do export "$e"; done
O=$R/gpurun_out/r3/pmc2_$TAG
mkdir -p $O
B="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline $BENCH_FLAGS"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_IFETCH --output-format csv -d $O/sq -- $B > /dev/null 2> $O/sq.log
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INST_LEVEL_LDS --output-format csv -d $O/ic -- $B > /dev/null 2> $O/ic.log
cd $R
python3 - $O <<'PY'
import csv,glob,collections,sys
O=sys.argv[1]
for sub in ('sq','ic'):
    fs=glob.glob(O+'/%s/*/*counter_collection.csv'%sub)+glob.glob(O+'/%s/*counter_collection.csv'%sub)
    if not fs: print('no csv for',sub); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(fs[0])):
        k=r['Kernel_Name'][:64]
        if 'persist' not in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k][r['Counter_Name']]+=1
    for k in agg:
        print(sub, k, {c:round(v/n[k][c]) for c,v in agg[k].items()})
PY
