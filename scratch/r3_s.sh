#!/bin/bash
# old (HEAD) library vs the working tree's, same box, alternating processes
set -o pipefail
mkdir -p gpurun_out; : > gpurun_out/r3s_ab.log
for rnd in 1 2; do
  for which in old new; do
    if [ $which = old ]; then export SBG_HIP_LIBRARY=$PWD/scratch/libsbg_hip_old.so; else unset SBG_HIP_LIBRARY; fi
    echo "== $which (round $rnd)" >> gpurun_out/r3s_ab.log
    timeout -k 10 300 python scratch/kbench_ab.py k64:0 up2:0 wgrad:0 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3s_ab.log || exit 1
  done
done
cat gpurun_out/r3s_ab.log
