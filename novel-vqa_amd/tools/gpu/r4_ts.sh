cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
for v in 3 2; do
for mode in "" "--arch 2 --bf16"; do
for d in 32 43; do
echo "== kernel v$v mode '$mode' PB_DBG=$d PF_DBG=32"
NVQA_BWD_KERNEL=$v NVQA_PB_DBG=$d NVQA_PF_DBG=32 timeout -k 10 120 python bench.py --steps 4 --warmup 1 --blocks 1 --no-cpu-baseline --no-secondary --no-roofline $mode 2>&1 >/dev/null | grep "nvqa\] persistent"
done
done
done
