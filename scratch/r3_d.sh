#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3d_gputest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3d_gputest.log
timeout -k 10 500 python bench.py --kernel-breakdown > gpurun_out/r3d_bench.json 2> gpurun_out/r3d_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
r=json.load(open('gpurun_out/r3d_bench.json'))
print({k:r[k] for k in ('value','ms_per_step','ms_per_step_median','roofline','target_kernel','kernel_ms_per_step','sbg_kernel_time_frac_of_step','nonfinite_grads')})
print([ (s['value'], s['ms_per_step']) for s in r.get('secondary',[])])
PY
