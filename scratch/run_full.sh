set -e
python -m pytest tests -m gpu -x -q > gpurun_out/full_tests.log 2>&1 || { tail -30 gpurun_out/full_tests.log; exit 1; }
tail -2 gpurun_out/full_tests.log
timeout -k 10 400 python bench.py --no-cpu-baseline --kernel-breakdown > gpurun_out/full_bench.json 2> gpurun_out/full_bench_breakdown.log
cat gpurun_out/full_bench.json
