set -e
python -m pytest tests -m gpu -x -q > gpurun_out/s2_tests.log 2>&1 || { tail -30 gpurun_out/s2_tests.log; exit 1; }
tail -3 gpurun_out/s2_tests.log
for v in 0 1 2 3; do echo "== SBG_CONV_TILE=$v"; SBG_CONV_TILE=$v timeout -k 10 120 python scratch/kbench.py conv 2>&1 | grep "halo=False\|s2\|convT"; done > gpurun_out/s2_kbench_conv.log 2>&1
cat gpurun_out/s2_kbench_conv.log
timeout -k 10 120 python scratch/kbench.py wgrad > gpurun_out/s2_kbench_wgrad.log 2>&1; cat gpurun_out/s2_kbench_wgrad.log
timeout -k 10 300 python bench.py --no-cpu-baseline --kernel-breakdown > gpurun_out/s2_bench.json 2> gpurun_out/s2_bench_breakdown.log
cat gpurun_out/s2_bench.json
