#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python scratch/halo_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3m_stamps.log; tail -45 gpurun_out/r3m_stamps.log
timeout -k 10 300 python scratch/kbench_ab.py k64:0,2 > gpurun_out/r3m_ab.log 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids gpurun_out/r3m_ab.log | tail -8
