#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python scratch/kbench_ab.py halo:0,2,34,66,3 up2:0,8 wgrad:0,16 gather:0,4 > gpurun_out/r3c_ab.log 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids gpurun_out/r3c_ab.log
