python -m pytest tests/test_gpu_arch1.py tests/test_gpu_arch2.py tests/test_gpu_fullsize.py tests/test_gpu_edge.py tests/test_gpu_variants.py -m gpu -q 2>&1 | tail -3
for r in "" "--ragged"; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline $r 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$r', d['ms_per_step'], d['kernel_ms_per_step'])"
done
