#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python scratch/kbench_ab.py haloepi:0,32,64 > gpurun_out/r3g_ab.log 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids gpurun_out/r3g_ab.log
