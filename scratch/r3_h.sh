#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python scratch/kbench_ab.py halo:0,128,256 haloepi:0,128,256 > gpurun_out/r3h_ab.log 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids gpurun_out/r3h_ab.log
