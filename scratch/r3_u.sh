#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -x -q -m gpu -k "grouped or style_bank or cat0" > gpurun_out/r3u_pytest_new.log 2>&1; rc=$?; echo "new tests rc=$rc"; tail -15 gpurun_out/r3u_pytest_new.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3u_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r3u_pytest.log
[ $rc -eq 0 ] || exit 1
for rnd in 1 2; do
  for sb in 0 1; do
    echo "== SBG_STYLE_BANK=$sb (round $rnd)"
    SBG_STYLE_BANK=$sb timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_median'], d['sbg_kernel_time_frac_of_step'])" || exit 1
  done
done
