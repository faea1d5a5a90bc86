#!/bin/bash
for bca in 64 128; do
  SBG_WGRAD_BCA=$bca timeout -k 10 300 python bench.py --workload big_gan --steps 8 --warmup 3 --kernel-breakdown > gpurun_out/r2r_big_gan_$bca.json 2> gpurun_out/r2r_big_gan_$bca.log
  echo "BCA=$bca: $(cut -c1-120 gpurun_out/r2r_big_gan_$bca.json | sed 's/.*"value": //' | cut -c1-40)"; grep "conv_wgrad" gpurun_out/r2r_big_gan_$bca.log | head -4
done
SBG_WGRAD_BCA=128 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --kernel-breakdown > gpurun_out/r2r_bench_128.json 2> gpurun_out/r2r_bench_128.log; cut -c1-130 gpurun_out/r2r_bench_128.json; grep "conv_wgrad " gpurun_out/r2r_bench_128.log | head -8
python -m pytest tests/test_engine_gpu.py -m gpu -x -q 2>&1 | tail -2
