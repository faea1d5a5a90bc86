#!/bin/bash
# A/B of the halo kernel's early-start schedule on ONE box: conv parity tests, then the bench with the switch off / on / off / on
python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv" > gpurun_out/r3f_test.log 2>&1; tail -2 gpurun_out/r3f_test.log
for v in 0 1 0 1; do
  SBG_HALO_EARLY=$v python bench.py --no-cpu-baseline --kernel-breakdown > gpurun_out/r3f_bench_$v.json 2> gpurun_out/r3f_bench_$v.err
  echo "EARLY=$v $(python -c "import json;d=json.load(open('gpurun_out/r3f_bench_$v.json'));print(d['value'],d['ms_per_step'],d['roofline']['achieved'],d['target_kernel']['tflops'])")"
  grep -A4 "top launches" gpurun_out/r3f_bench_$v.err | tail -3
done
