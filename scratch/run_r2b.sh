#!/bin/bash
set -o pipefail
timeout -k 10 500 python bench.py --workload ffhq_sg2 --steps 4 --warmup 1 --kernel-breakdown > gpurun_out/r2b_bench_ffhq.json 2> gpurun_out/r2b_bench_ffhq.log || { echo "ffhq failed"; tail -5 gpurun_out/r2b_bench_ffhq.log; }
cut -c1-500 gpurun_out/r2b_bench_ffhq.json
bash profiles/collect.sh r02a > gpurun_out/r2b_collect.log 2>&1 || { echo "collect failed"; tail -20 gpurun_out/r2b_collect.log; }
tail -5 gpurun_out/r2b_collect.log
