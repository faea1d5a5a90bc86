#!/bin/bash
set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/r2s_tests.log 2>&1 || tail -30 gpurun_out/r2s_tests.log
tail -2 gpurun_out/r2s_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
