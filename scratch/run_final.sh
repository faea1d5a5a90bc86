#!/bin/bash
set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/final_tests.log 2>&1 || tail -30 gpurun_out/final_tests.log
tail -2 gpurun_out/final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/final_bench_default.json 2> gpurun_out/final_bench_default.err || tail -20 gpurun_out/final_bench_default.err
cut -c1-220 gpurun_out/final_bench_default.json
