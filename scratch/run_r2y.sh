#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 4 --warmup 2 --launch-log gpurun_out/r2y_launch.jsonl > gpurun_out/r2y_bench2.json 2>gpurun_out/r2y_bench2.err; cut -c1-200 gpurun_out/r2y_bench2.json
SBG_CONV_NO_UP2=1 timeout -k 10 300 python bench.py --steps 4 --warmup 2 > gpurun_out/r2y_bench3.json 2>gpurun_out/r2y_bench3.err; cut -c1-200 gpurun_out/r2y_bench3.json
