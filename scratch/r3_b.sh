#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python scratch/kbench_ab.py 0 1 2 3 --all > gpurun_out/r3b_ab.log 2>&1; echo "ab rc=$?"; cat gpurun_out/r3b_ab.log | grep -v amdgpu.ids
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r3b_gputest.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r3b_gputest.log
