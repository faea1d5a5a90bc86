#!/bin/bash
# sliced leading discriminator blocks (a merged Dmain pass of 128 at 256x256): tests, then bench with the D merge off / on on one box
python -m pytest tests/test_engine_gpu.py tests/test_reference_vectors_gpu.py -m gpu -x -q > gpurun_out/r3j_test.log 2>&1; tail -3 gpurun_out/r3j_test.log
for v in 0 1 0 1; do
  SBG_MERGE_D=$v python bench.py --no-cpu-baseline --kernel-breakdown > gpurun_out/r3j_bench_$v.json 2> gpurun_out/r3j_bench_$v.err
  echo "MERGE_D=$v $(python -c "import json;d=json.load(open('gpurun_out/r3j_bench_$v.json'));print(d['value'],d['ms_per_step'],d['ms_per_step_median'],d['sbg_kernel_time_frac_of_step'])")"
done
