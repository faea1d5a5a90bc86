#!/bin/bash
for t in 512 2048; do
  export SBG_WGRAD_TARGET=$t
  timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/wt_bench_$t.json 2> /dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/wt_bench_$t.json').read().strip().splitlines()[-1]); print($t, d['value'], d['ms_per_step'], d['kernel_ms_per_step']['conv_wgrad'], d['kernel_ms_per_step']['wgrad_reduce'])"
done
