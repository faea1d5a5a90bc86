#!/bin/bash
# precision variants of the headline workload as secondary lines: the reference recipe's num_fp16_res = 4, and fp32 storage everywhere
mkdir -p gpurun_out
: > gpurun_out/r02e_precision.jsonl
for args in "--num-fp16-res 4 --steps 8 --warmup 3" "--num-fp16-res 0 --steps 4 --warmup 2"; do
  timeout -k 10 500 python bench.py $args --no-cpu-baseline 2> gpurun_out/r02e_prec_err.log | tail -1 >> gpurun_out/r02e_precision.jsonl || { tail -5 gpurun_out/r02e_prec_err.log; }
  tail -1 gpurun_out/r02e_precision.jsonl | cut -c1-160
done
