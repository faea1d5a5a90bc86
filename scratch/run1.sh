python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout -k 10 400 python bench.py --no-cpu-baseline --kernel-breakdown > gpurun_out/s15_bench.json 2> gpurun_out/s15_bench_breakdown.log; cat gpurun_out/s15_bench.json | cut -c1-500; grep "(512, 51" gpurun_out/s15_bench_breakdown.log
