#!/bin/bash
# profiles of the merged-rounds build (tag r02e) + the secondary workload lines
mkdir -p gpurun_out
bash profiles/collect.sh r02e > gpurun_out/collect_r02e.log 2>&1; echo "collect rc=$?"; tail -4 gpurun_out/collect_r02e.log
: > gpurun_out/r02e_secondary.jsonl
for args in "--workload ffhq_sg2 --steps 16 --warmup 2" "--workload big_gan --steps 8 --warmup 3" "--workload sg2attent --steps 8 --warmup 3" "--workload sg2attent --res 256 --steps 4 --warmup 2" "--ada 0.5 --steps 8 --warmup 3"; do
  timeout -k 10 400 python bench.py $args --no-cpu-baseline 2> gpurun_out/r02e_err.log | tail -1 >> gpurun_out/r02e_secondary.jsonl || { tail -5 gpurun_out/r02e_err.log; }
  tail -1 gpurun_out/r02e_secondary.jsonl | cut -c1-150
done
