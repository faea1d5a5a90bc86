#!/bin/bash
set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/r2m_tests.log 2>&1 || tail -30 gpurun_out/r2m_tests.log
tail -2 gpurun_out/r2m_tests.log
python bench.py --steps 8 --warmup 3 --kernel-breakdown > gpurun_out/r2m_bench.json 2> gpurun_out/r2m_bench_breakdown.log || tail -20 gpurun_out/r2m_bench_breakdown.log
cut -c1-200 gpurun_out/r2m_bench.json
python - <<'PY'
import json; d=json.load(open('gpurun_out/r2m_bench.json')); print(d['kernel_ms_per_step'], d['sbg_kernel_time_frac_of_step'], d['ms_per_step_median'])
PY
