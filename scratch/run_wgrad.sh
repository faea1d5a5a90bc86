#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_networks_gpu.py -x -q -k "conv or network or forward or gradients or synthesis" > gpurun_out/conv_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/conv_tests.log
timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --kernel-breakdown > gpurun_out/full_bench.json 2> gpurun_out/full_bench_breakdown.log; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/full_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
grep "conv_wgrad " gpurun_out/full_bench_breakdown.log | head -14
