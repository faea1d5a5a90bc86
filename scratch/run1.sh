timeout -k 10 120 python scratch/k64_check.py > gpurun_out/s13_check.log 2>&1 || { tail -40 gpurun_out/s13_check.log; exit 1; }; tail -1 gpurun_out/s13_check.log
for v in 1 0; do echo "== LD=$v"; SBG_K64_LD=$v timeout -k 10 120 python scratch/kbench.py conv2 2>&1 | grep "conv3x3"; done > gpurun_out/s13_ld.log 2>&1
cat gpurun_out/s13_ld.log
python -m pytest tests -m gpu -x -q -k "ops or networks" 2>&1 | tail -2
timeout -k 10 400 python bench.py --no-cpu-baseline --kernel-breakdown > gpurun_out/s13_bench.json 2> gpurun_out/s13_bench_breakdown.log; cat gpurun_out/s13_bench.json | cut -c1-1500
