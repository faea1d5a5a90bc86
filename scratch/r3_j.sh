#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python scratch/kbench_ab.py halo:64,0,256 haloepi:64,0 > gpurun_out/r3j_ab.log 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids gpurun_out/r3j_ab.log | tail -30
