#!/bin/bash
set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/r2q_tests.log 2>&1 || tail -30 gpurun_out/r2q_tests.log
tail -2 gpurun_out/r2q_tests.log
for wl in big_gan sg2attent; do
  timeout -k 10 300 python bench.py --workload $wl --steps 8 --warmup 3 --kernel-breakdown > gpurun_out/r2q_bench_$wl.json 2> gpurun_out/r2q_bench_$wl.log || { echo "$wl failed"; tail -20 gpurun_out/r2q_bench_$wl.log; }
  cut -c1-160 gpurun_out/r2q_bench_$wl.json
done
python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r2q_bench.json 2>/dev/null; cut -c1-160 gpurun_out/r2q_bench.json
