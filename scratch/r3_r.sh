#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3r_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r3r_pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python scratch/kbench_ab.py k64:0 up2:0 wgrad:0 halo:0 > gpurun_out/r3r_ab.log 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids gpurun_out/r3r_ab.log | tail -20
timeout -k 10 400 python bench.py --no-secondary > gpurun_out/r3r_bench.json 2> gpurun_out/r3r_bench.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/r3r_bench.json
