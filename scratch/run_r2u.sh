#!/bin/bash
python scratch/fp32_passes.py 2>&1 | tail -5
SBG_FP32_PASSES=3 python scratch/fp32_passes.py 2>&1 | tail -5
SBG_FP32_PASSES=3 python -m pytest tests -m gpu -q > gpurun_out/r2u_tests_3pass.log 2>&1; tail -15 gpurun_out/r2u_tests_3pass.log
SBG_FP32_PASSES=3 timeout -k 10 300 python bench.py --workload big_gan --no-cpu-baseline 2>/dev/null | cut -c1-150
