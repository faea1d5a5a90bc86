#!/bin/bash
for t in 256 16; do
  export SBG_PHASE_MIN_TILES=$t
  timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "conv2d_shapes or resample or up_synthesis" > gpurun_out/ph_tests_$t.log 2>&1; echo "t=$t tests rc=$?"; tail -1 gpurun_out/ph_tests_$t.log
  timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/ph_bench_$t.json 2> /dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/ph_bench_$t.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step']['conv_igemm'])"
done
