cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
for cfg in "NVQA_PF_DBG=256" "NVQA_PF_DBG=0"; do
echo "== $cfg"
env $cfg timeout -k 10 300 python -m pytest tests/test_gpu_parity_r2.py -q -m gpu -k "headline or persistent_forward_lstm" -rf 2>&1 | grep -E "passed|failed|AssertionError: \(|Error|FAILED" | head -8
done
