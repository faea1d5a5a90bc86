#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3l_gputest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3l_gputest.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --kernel-breakdown > gpurun_out/r3l_bench.json 2> gpurun_out/r3l_bench.err || exit 1
python - <<'PY'
import json
r = json.load(open('gpurun_out/r3l_bench.json'))
print(r['value'], 'img/s', r['ms_per_step'], 'ms', r['kernel_ms_per_step'], 'target', r['target_kernel']['tflops'], flush=True)
PY
grep -E "^  conv_igemm.*(9064256|1128256)" gpurun_out/r3l_bench.err | head -10
