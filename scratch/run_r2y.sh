#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --workload ffhq_sg2 --steps 4 --warmup 2 --no-cpu-baseline --launch-log gpurun_out/r2y_launch_ffhq.jsonl > gpurun_out/r2y_ffhq.json 2>gpurun_out/r2y_ffhq.err; cut -c1-200 gpurun_out/r2y_ffhq.json
