#!/bin/bash
# three builds of the library on one box, alternating processes: old2 = e912c19 (before the loader changes), old = HEAD, new = working tree
set -o pipefail
mkdir -p gpurun_out; : > gpurun_out/r3t_ab.log
for rnd in 1 2; do
  for which in old2 old new; do
    if [ $which = new ]; then unset SBG_HIP_LIBRARY; else export SBG_HIP_LIBRARY=$PWD/scratch/libsbg_hip_$which.so; fi
    echo "== $which (round $rnd)" >> gpurun_out/r3t_ab.log
    timeout -k 10 300 python scratch/kbench_ab.py halo:0 haloepi:0 k64:0 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3t_ab.log || exit 1
    timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | cut -c1-200 >> gpurun_out/r3t_ab.log || exit 1
  done
done
cat gpurun_out/r3t_ab.log
