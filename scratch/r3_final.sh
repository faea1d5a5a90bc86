#!/bin/bash
# final evidence of the round on one box: CPU-side checks are done in the container; here the GPU suite, then the profile collection (r03c)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3c_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r3c_pytest.log
[ $rc -eq 0 ] || exit 1
bash scratch/collect_r03.sh ${1:-r03d}
