#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2y_tests.log 2>&1; tail -3 gpurun_out/r2y_tests.log
timeout -k 10 300 python bench.py --steps 8 --warmup 3 > gpurun_out/r2y_bench.json 2>gpurun_out/r2y_bench.err; cut -c1-200 gpurun_out/r2y_bench.json
