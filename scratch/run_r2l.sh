#!/bin/bash
set -o pipefail
bash scratch/run_dist2.sh
timeout -k 10 500 python bench.py --workload ffhq_sg2 --steps 4 --warmup 1 > gpurun_out/r2l_bench_ffhq.json 2> gpurun_out/r2l_bench_ffhq.log || { echo "ffhq failed"; tail -5 gpurun_out/r2l_bench_ffhq.log; }
cut -c1-160 gpurun_out/r2l_bench_ffhq.json
python -m pytest tests/test_ops_gpu.py tests/test_networks_gpu.py -m gpu -x -q 2>&1 | tail -2
