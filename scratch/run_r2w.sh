#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 200 python scratch/thin_check.py 2>&1 | tail -8
echo "--- TP8"
SBG_THIN_TP4=0 timeout -k 10 200 python scratch/thin_check.py 2>&1 | tail -7
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_reference_vectors_gpu.py -x -q -m gpu > gpurun_out/r2w_tests.log 2>&1; tail -3 gpurun_out/r2w_tests.log
