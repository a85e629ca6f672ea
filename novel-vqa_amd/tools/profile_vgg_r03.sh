#!/bin/bash
# Round-3 extractor profile (GPU box, from the repo root): per-layer durations, matrix-pipe busy cycles, wave-cycle shares and
# HBM traffic of tools/bench_vgg.py.  usage: profile_vgg_r03.sh [tag [batch [bench_vgg flags]]]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_vgg_r03${1:+_$1}
mkdir -p $O
B="python3 $R/novel-vqa_amd/tools/bench_vgg.py ${2:-32} 2 $3"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/bench_vgg.json 2> $O/stats.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- $B > /dev/null 2> $O/mfma.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq -- $B > /dev/null 2> $O/sq.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > /dev/null 2> $O/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > /dev/null 2> $O/write.log
cd $R
cp $(ls $O/stats/*/*kernel_stats.csv $O/stats/*kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats.csv
python3 novel-vqa_amd/tools/pmc_vgg_layers.py $O/stats $O/mfma $O/sq $O/fetch $O/write $O/layers.json > $O/layers.txt
cat $O/layers.txt
cat $O/bench_vgg.json
