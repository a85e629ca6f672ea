#!/bin/bash
# Round-4 profile set (run on the GPU box from the repo root): kernel statistics, HBM traffic (FETCH_SIZE / WRITE_SIZE in
# separate passes, MI355X_MICROARCH.md), matrix-pipe busy cycles, wave-cycle shares.  Results -> gpurun_out/prof_r04/,
# summaries -> profiles/r04_*.  The program after `--` is python3 itself (no env / bash hop under rocprofv3).
# usage: profile_r04.sh [tag [bench flags]]   e.g.  profile_r04.sh bf16_arch2 "--arch 2 --bf16"
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04${1:+_$1}
mkdir -p $O
B="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary $2"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/bench_under_rocprof.json 2> $O/stats.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > /dev/null 2> $O/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > /dev/null 2> $O/write.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- $B > /dev/null 2> $O/mfma.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES --output-format csv -d $O/sq -- $B > /dev/null 2> $O/sq.log
cd $R
python3 novel-vqa_amd/tools/pmc_traffic.py $O/fetch $O/write $O/traffic.json > /dev/null
python3 novel-vqa_amd/tools/pmc_mfma.py $O/mfma $O/mfma_util.json
python3 novel-vqa_amd/tools/pmc_sq.py $O/sq $O/sq_counters.json
cp $(ls $O/stats/*/*kernel_stats.csv $O/stats/*kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats.csv
ls -la $O
tail -c 600 $O/bench_under_rocprof.json
