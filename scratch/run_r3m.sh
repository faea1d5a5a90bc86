#!/bin/bash
# end-of-round record of the current build: smoke, profiles (tag r02i), default bench (with the CPU baseline), secondary + precision lines
mkdir -p gpurun_out
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
bash profiles/collect.sh r02i > gpurun_out/collect_r02i.log 2>&1; echo "collect rc=$?"; tail -3 gpurun_out/collect_r02i.log
timeout -k 10 500 python bench.py > gpurun_out/r02i_bench_default.json 2> gpurun_out/r02i_bench_default.err; echo "default bench rc=$?"; cut -c1-200 gpurun_out/r02i_bench_default.json
: > gpurun_out/r02i_secondary.jsonl
for args in "--workload ffhq_sg2 --steps 16 --warmup 2" "--workload big_gan --steps 8 --warmup 3" "--workload sg2attent --steps 8 --warmup 3" "--workload sg2attent --res 256 --steps 4 --warmup 2" "--ada 0.5 --steps 8 --warmup 3" "--num-fp16-res 4 --steps 8 --warmup 3" "--num-fp16-res 0 --steps 4 --warmup 2"; do
  timeout -k 10 500 python bench.py $args --no-cpu-baseline 2> gpurun_out/r02i_err.log | tail -1 >> gpurun_out/r02i_secondary.jsonl || { tail -5 gpurun_out/r02i_err.log; }
  tail -1 gpurun_out/r02i_secondary.jsonl | cut -c1-150
done
