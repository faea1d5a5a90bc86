#!/bin/bash
# round 3, first GPU call: gpu tests, default bench (with secondary + cpu baseline), launcher failure + gloo rehearsal
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a_gputest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3a_gputest.log
tail -5 gpurun_out/r3a_gputest.log
timeout -k 10 400 python bench.py --kernel-breakdown > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err; echo "bench rc=$?"
cat gpurun_out/r3a_bench.json | head -c 6000
python bench.py --gpus 2 > gpurun_out/r3a_bench2_refused.json 2> gpurun_out/r3a_bench2_refused.err; echo "bench --gpus 2 (one device) rc=$?"; tail -2 gpurun_out/r3a_bench2_refused.err
SBG_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --batch 16 --batch-gpu 8 > gpurun_out/r3a_bench2_gloo.json 2> gpurun_out/r3a_bench2_gloo.err; echo "gloo rehearsal rc=$?"
cat gpurun_out/r3a_bench2_gloo.json | head -c 3000
