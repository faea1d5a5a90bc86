#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 python scratch/kbench_ab.py halo:256,0 haloepi:256,0 > gpurun_out/r3i_ab.log 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids gpurun_out/r3i_ab.log | tail -30
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_networks_gpu.py tests/test_reference_vectors_gpu.py tests/test_engine_gpu.py -m gpu -q -x > gpurun_out/r3i_gputest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3i_gputest.log
