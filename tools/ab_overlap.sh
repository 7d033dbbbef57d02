set -e
timeout -k 10 300 python -m pytest tests/test_gpu_entry_points.py -x -q -m gpu -k "overlapped" 2>&1 | tail -3
for i in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --overlap > gpurun_out/bench_r03_ov.json 2>/dev/null
  python tools/print_bench.py gpurun_out/bench_r03_ov.json overlap
  timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/bench_r03_noov.json 2>/dev/null
  python tools/print_bench.py gpurun_out/bench_r03_noov.json no-overlap
done
