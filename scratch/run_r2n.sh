#!/bin/bash
set -o pipefail
python -m pytest tests/test_ops_gpu.py tests/test_networks_gpu.py tests/test_reference_vectors_gpu.py -m gpu -x -q > gpurun_out/r2n_tests.log 2>&1 || tail -30 gpurun_out/r2n_tests.log
tail -2 gpurun_out/r2n_tests.log
python bench.py --steps 8 --warmup 3 --kernel-breakdown > gpurun_out/r2n_bench.json 2> gpurun_out/r2n_bench_breakdown.log || tail -20 gpurun_out/r2n_bench_breakdown.log
cut -c1-200 gpurun_out/r2n_bench.json
python - <<'PY'
import json; d=json.load(open('gpurun_out/r2n_bench.json')); print(d['kernel_ms_per_step'], d['sbg_kernel_time_frac_of_step'], d['ms_per_step_median'])
PY
python bench.py --steps 8 --warmup 3 > gpurun_out/r2n_bench2.json 2>/dev/null; cut -c1-160 gpurun_out/r2n_bench2.json
