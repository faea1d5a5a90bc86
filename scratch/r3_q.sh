#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python scratch/halo_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3q_stamps.log; tail -45 gpurun_out/r3q_stamps.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3q_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3q_pytest.log
timeout -k 10 400 python bench.py --no-secondary > gpurun_out/r3q_bench.json 2> gpurun_out/r3q_bench.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/r3q_bench.json
