#!/bin/bash
set -o pipefail
mkdir -p gpurun_out


for e in 0 32 0 32; do
  SBG_EXPERIMENT=$e timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r3f_bench_$e.json 2> gpurun_out/r3f_bench_$e.err || exit 1
  python - $e <<'PY'
import json, sys
e = sys.argv[1]
r = json.load(open(f'gpurun_out/r3f_bench_{e}.json'))
print('exp', e, r['value'], 'img/s', r['ms_per_step'], 'ms', {k: r['kernel_ms_per_step'][k] for k in ('conv_igemm', 'conv_wgrad')}, 'target', r['target_kernel']['tflops'], flush=True)
PY
done
