#!/bin/bash
# VGG-16 fc7 extractor: kernel statistics + HBM traffic (PMC) of tools/bench_vgg.py (batch 32, full 224x224 network)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_vgg_r02
mkdir -p $O
B="python3 $R/novel-vqa_amd/tools/bench_vgg.py 32 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/bench_vgg.json 2> $O/stats.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > /dev/null 2> $O/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > /dev/null 2> $O/write.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- $B > /dev/null 2> $O/mfma.log
cd $R
cp $(ls $O/stats/*/*kernel_stats.csv $O/stats/*kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats.csv
python3 - <<PY
import csv, glob, json, collections
O="$O"
def load(d):
    f=(glob.glob(d+"/*/*counter_collection.csv")+glob.glob(d+"/*counter_collection.csv"))[0]
    return list(csv.DictReader(open(f)))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for d,c in (("fetch","FETCH_SIZE"),("write","WRITE_SIZE")):
    for r in load(O+"/"+d):
        if r["Counter_Name"]==c:
            k=r["Kernel_Name"].split("(")[0][:110]
            agg[k][c]+=float(r["Counter_Value"]); 
            if c=="FETCH_SIZE": cnt[k]+=1
m=collections.defaultdict(lambda: collections.defaultdict(float))
for r in load(O+"/mfma"):
    k=r["Kernel_Name"].split("(")[0][:110]
    m[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="SQ_VALU_MFMA_BUSY_CYCLES": m[k]["ns"]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
out={}
for k in agg:
    n=max(cnt[k],1)
    out[k]={"launches":cnt[k],"hbm_bytes_per_launch":round((2*agg[k]["FETCH_SIZE"]+agg[k]["WRITE_SIZE"])*1024/n),
            "mfma_busy_frac": round(m[k]["SQ_VALU_MFMA_BUSY_CYCLES"]/(m[k]["ns"]*1e-9*2.38e9*1024),4) if m[k]["ns"] else None}
json.dump(out,open(O+"/vgg_pmc.json","w"),indent=1)
print(json.dumps(out,indent=1)[:1500])
PY
cat $O/bench_vgg.json
