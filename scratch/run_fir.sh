#!/bin/bash
mkdir -p gpurun_out
for rpw in 4; do
  export SBG_FIR_RPW=$rpw
  timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "upfirdn or up_synthesis" > gpurun_out/fir_tests_$rpw.log 2>&1; echo "rpw=$rpw tests rc=$?"; tail -1 gpurun_out/fir_tests_$rpw.log
  timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --kernel-breakdown > gpurun_out/fir_bench_$rpw.json 2> gpurun_out/fir_bench_$rpw.log; echo "bench rc=$?"
  python -c "
import json; d=json.loads(open('gpurun_out/fir_bench_$rpw.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step']['upfirdn2d'])"
  grep "upfirdn2d " gpurun_out/fir_bench_$rpw.log | head -6
done
