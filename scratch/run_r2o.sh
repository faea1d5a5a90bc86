#!/bin/bash
set -o pipefail
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2o_smoke.log 2>&1 || { echo "smoke failed"; tail -20 gpurun_out/r2o_smoke.log; }
tail -1 gpurun_out/r2o_smoke.log
python bench.py > gpurun_out/r2o_bench_default.json 2> gpurun_out/r2o_bench_default.err || tail -20 gpurun_out/r2o_bench_default.err
cut -c1-300 gpurun_out/r2o_bench_default.json
bash profiles/collect.sh r02c > gpurun_out/r2o_collect.log 2>&1 || { echo "collect failed"; tail -20 gpurun_out/r2o_collect.log; }
grep -E "pass done|traffic" gpurun_out/r2o_collect.log
