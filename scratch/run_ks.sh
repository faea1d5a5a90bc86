#!/bin/bash
for cfg in "128 256" "320 512" "128 512"; do
  set -- $cfg
  export SBG_KSPLIT_MAX_TILES=$1 SBG_KSPLIT_TARGET=$2
  timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/ks_bench.json 2> /dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/ks_bench.json').read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'], d['kernel_ms_per_step']['conv_igemm'])"
done
