#!/bin/bash
set -o pipefail
python -m pytest tests/test_engine_gpu.py -m gpu -x -q > gpurun_out/r2c_tests.log 2>&1 || { tail -30 gpurun_out/r2c_tests.log; }
tail -2 gpurun_out/r2c_tests.log
python scratch/op_profile.py > gpurun_out/r2c_op_profile.log 2>&1 || tail -5 gpurun_out/r2c_op_profile.log
bash profiles/collect.sh r02a > gpurun_out/r2c_collect.log 2>&1 || { echo "collect failed"; tail -20 gpurun_out/r2c_collect.log; }
tail -5 gpurun_out/r2c_collect.log
