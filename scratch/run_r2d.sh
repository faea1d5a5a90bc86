#!/bin/bash
set -o pipefail
python scratch/determinism.py > gpurun_out/r2d_determinism.log 2>&1; tail -12 gpurun_out/r2d_determinism.log
python -m pytest tests -m gpu -x -q --deselect tests/test_engine_gpu.py::test_snapshot_resume_continuity_on_device > gpurun_out/r2d_tests.log 2>&1 || tail -30 gpurun_out/r2d_tests.log
tail -2 gpurun_out/r2d_tests.log
python bench.py --steps 8 --warmup 3 --kernel-breakdown > gpurun_out/r2d_bench.json 2> gpurun_out/r2d_bench_breakdown.log || tail -20 gpurun_out/r2d_bench_breakdown.log
cut -c1-300 gpurun_out/r2d_bench.json
bash profiles/collect.sh r02a > gpurun_out/r2d_collect.log 2>&1 || { echo "collect failed"; tail -20 gpurun_out/r2d_collect.log; }
tail -6 gpurun_out/r2d_collect.log
