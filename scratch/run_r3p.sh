#!/bin/bash
# chained backward heads in synthesis blocks: tests, then the bench with the chaining off / on on one box
python -m pytest tests/test_networks_gpu.py -m gpu -x -q -k "chained" > gpurun_out/r3p_test0.log 2>&1; tail -15 gpurun_out/r3p_test0.log
python -m pytest tests -m gpu -x -q > gpurun_out/r3p_test.log 2>&1; tail -3 gpurun_out/r3p_test.log
for v in 0 1 0 1; do
  SBG_CHAIN_HEADS=$v python bench.py --no-cpu-baseline --kernel-breakdown > gpurun_out/r3p_bench_$v.json 2> gpurun_out/r3p_bench_$v.err
  echo "CHAIN=$v $(python -c "import json;d=json.load(open('gpurun_out/r3p_bench_$v.json'));print(d['value'],d['ms_per_step'],d['ms_per_step_median'],d['kernel_ms_per_step'])")"
done
