#!/bin/bash
# round-2 first measurement: full GPU tests, headline bench with breakdown, secondary workloads, profiles
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2a_tests.log 2>&1 || { tail -30 gpurun_out/r2a_tests.log; exit 1; }
tail -3 gpurun_out/r2a_tests.log
python bench.py --steps 8 --warmup 3 --kernel-breakdown > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench_breakdown.log || { tail -20 gpurun_out/r2a_bench_breakdown.log; exit 1; }
cat gpurun_out/r2a_bench.json | cut -c1-600
for wl in sg2attent big_gan; do
  timeout -k 10 300 python bench.py --workload $wl --steps 8 --warmup 3 --kernel-breakdown > gpurun_out/r2a_bench_$wl.json 2> gpurun_out/r2a_bench_$wl.log || { echo "$wl failed"; tail -20 gpurun_out/r2a_bench_$wl.log; }
  cut -c1-400 gpurun_out/r2a_bench_$wl.json
done
timeout -k 10 400 python bench.py --workload ffhq_sg2 --steps 4 --warmup 1 --kernel-breakdown > gpurun_out/r2a_bench_ffhq.json 2> gpurun_out/r2a_bench_ffhq.log || { echo "ffhq failed"; tail -20 gpurun_out/r2a_bench_ffhq.log; }
cut -c1-400 gpurun_out/r2a_bench_ffhq.json
bash profiles/collect.sh r02a > gpurun_out/r2a_collect.log 2>&1 || { echo "collect failed"; tail -20 gpurun_out/r2a_collect.log; }
tail -5 gpurun_out/r2a_collect.log
