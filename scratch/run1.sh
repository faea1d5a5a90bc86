python -m pytest tests -m gpu -x -q -k "conv" 2>&1 | tail -2
timeout -k 10 400 python bench.py --no-cpu-baseline --kernel-breakdown > gpurun_out/s16_bench.json 2> gpurun_out/s16_bench_breakdown.log; cat gpurun_out/s16_bench.json | cut -c1-330; grep -A5 '"wgrad_reduce"' gpurun_out/s16_bench_breakdown.log | head -6
