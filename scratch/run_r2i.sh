#!/bin/bash
set -o pipefail
timeout -k 10 300 python scratch/thin_check.py > gpurun_out/r2i_thin.log 2>&1; tail -9 gpurun_out/r2i_thin.log
python -m pytest tests/test_engine_gpu.py tests/test_ops_gpu.py tests/test_networks_gpu.py -m gpu -x -q > gpurun_out/r2i_tests.log 2>&1 || tail -30 gpurun_out/r2i_tests.log
tail -2 gpurun_out/r2i_tests.log
