#!/bin/bash
# secondary workloads with the current build (one JSON line each)
mkdir -p gpurun_out
: > gpurun_out/r2s_secondary.jsonl
for args in "--workload ffhq_sg2 --steps 16 --warmup 2" "--workload big_gan --steps 8 --warmup 3" "--workload sg2attent --steps 8 --warmup 3" "--workload sg2attent --res 256 --steps 4 --warmup 2" "--ada 0.5 --steps 8 --warmup 3"; do
  timeout -k 10 400 python bench.py $args --no-cpu-baseline 2> gpurun_out/r2s_err.log | tail -1 >> gpurun_out/r2s_secondary.jsonl || { tail -5 gpurun_out/r2s_err.log; }
  tail -1 gpurun_out/r2s_secondary.jsonl | cut -c1-150
done
