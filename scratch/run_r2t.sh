#!/bin/bash
mkdir -p gpurun_out/final
for spec in "big_gan" "sg2attent" "sg2attent --res 256" "ffhq_sg2 --steps 16 --warmup 2"; do
  tag=$(echo $spec | tr ' ' '_' | tr -d '-')
  timeout -k 10 500 python bench.py --workload $spec --no-cpu-baseline > gpurun_out/final/bench_$tag.json 2> gpurun_out/final/bench_$tag.err || { echo "$spec failed"; tail -5 gpurun_out/final/bench_$tag.err; }
  cut -c1-170 gpurun_out/final/bench_$tag.json
done
python bench.py --ada 0.5 --no-cpu-baseline > gpurun_out/final/bench_ada.json 2>/dev/null; cut -c1-170 gpurun_out/final/bench_ada.json
