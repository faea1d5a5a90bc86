#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 120 python scratch/wgrad_abl.py 2>&1 | grep TF
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_reference_vectors_gpu.py -x -q -m gpu > gpurun_out/r2z_tests.log 2>&1; tail -3 gpurun_out/r2z_tests.log
