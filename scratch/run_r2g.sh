#!/bin/bash
set -o pipefail
python -m pytest tests/test_engine_gpu.py -m gpu -x -q > gpurun_out/r2g_tests.log 2>&1 || tail -30 gpurun_out/r2g_tests.log
tail -2 gpurun_out/r2g_tests.log
bash profiles/collect.sh r02b > gpurun_out/r2g_collect.log 2>&1 || { echo "collect failed"; tail -20 gpurun_out/r2g_collect.log; }
tail -6 gpurun_out/r2g_collect.log
for wl in sg2attent big_gan; do
  timeout -k 10 300 python bench.py --workload $wl --steps 8 --warmup 3 --kernel-breakdown > gpurun_out/r2g_bench_$wl.json 2> gpurun_out/r2g_bench_$wl.log || { echo "$wl failed"; tail -20 gpurun_out/r2g_bench_$wl.log; }
  cut -c1-160 gpurun_out/r2g_bench_$wl.json
done
