#!/bin/bash
mkdir -p gpurun_out
for bca in 64 128; do for ns in 2 3 4 5 6; do echo "BCA $bca NSTAGE $ns"; SBG_WGRAD_BCA=$bca SBG_WGRAD_NSTAGE=$ns timeout -k 10 120 python scratch/wgrad_abl.py 2>&1 | grep TF; done; done
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu > gpurun_out/r2z_tests.log 2>&1; tail -3 gpurun_out/r2z_tests.log
