#!/bin/bash
# secondary workloads after Discriminator.pass_plan (ffhq_sg2 at 1024x1024 now slices its 1024 / 512 blocks)
mkdir -p gpurun_out
: > gpurun_out/r3l_secondary.jsonl
for args in "--workload ffhq_sg2 --steps 16 --warmup 2" "--workload big_gan --steps 8 --warmup 3" "--workload sg2attent --steps 8 --warmup 3" "--ada 0.5 --steps 8 --warmup 3"; do
  timeout -k 10 400 python bench.py $args --no-cpu-baseline 2> gpurun_out/r3l_err.log | tail -1 >> gpurun_out/r3l_secondary.jsonl || { tail -5 gpurun_out/r3l_err.log; }
  tail -1 gpurun_out/r3l_secondary.jsonl | cut -c1-150
done
