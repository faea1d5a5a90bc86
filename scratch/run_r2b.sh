#!/bin/bash
set -o pipefail
python -m pytest tests/test_biggan_gpu.py tests/test_reference_vectors_gpu.py -m gpu -x -q > gpurun_out/r2b_tests.log 2>&1 || { tail -30 gpurun_out/r2b_tests.log; exit 1; }
tail -2 gpurun_out/r2b_tests.log
timeout -k 10 500 python bench.py --workload ffhq_sg2 --steps 4 --warmup 1 --kernel-breakdown > gpurun_out/r2b_bench_ffhq.json 2> gpurun_out/r2b_bench_ffhq.log || { echo "ffhq failed"; tail -5 gpurun_out/r2b_bench_ffhq.log; }
cut -c1-500 gpurun_out/r2b_bench_ffhq.json
bash profiles/collect.sh r02a > gpurun_out/r2b_collect.log 2>&1 || { echo "collect failed"; tail -20 gpurun_out/r2b_collect.log; }
tail -5 gpurun_out/r2b_collect.log
