#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2y_tests.log 2>&1; tail -5 gpurun_out/r2y_tests.log
timeout -k 10 300 python bench.py --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r2y_m1.json 2>gpurun_out/r2y_m1.err; cut -c1-200 gpurun_out/r2y_m1.json
SBG_MERGE_D=0 timeout -k 10 300 python bench.py --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r2y_m0.json 2>gpurun_out/r2y_m0.err; cut -c1-200 gpurun_out/r2y_m0.json
