#!/bin/bash
set -o pipefail
timeout -k 10 500 python bench.py --workload ffhq_sg2 --steps 4 --warmup 1 --kernel-breakdown > gpurun_out/r2j_bench_ffhq.json 2> gpurun_out/r2j_bench_ffhq.log || { echo "ffhq failed"; tail -5 gpurun_out/r2j_bench_ffhq.log; }
cut -c1-200 gpurun_out/r2j_bench_ffhq.json
python - <<'PY'
import json; d=json.load(open('gpurun_out/r2j_bench_ffhq.json')); print(d['kernel_ms_per_step'], d['sbg_kernel_time_frac_of_step'], d['roofline'])
PY
grep -A22 "top launches" gpurun_out/r2j_bench_ffhq.log | head -24
python -m pytest tests -m gpu -x -q > gpurun_out/r2j_tests.log 2>&1 || tail -30 gpurun_out/r2j_tests.log
tail -2 gpurun_out/r2j_tests.log
