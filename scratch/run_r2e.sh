#!/bin/bash
set -o pipefail
python scratch/fir_bench.py > gpurun_out/r2e_fir.log 2>&1; tail -18 gpurun_out/r2e_fir.log
SBG_FIR_RPW=2 python scratch/fir_bench.py > gpurun_out/r2e_fir_tile.log 2>&1; tail -9 gpurun_out/r2e_fir_tile.log
python scratch/determinism.py > gpurun_out/r2e_determinism.log 2>&1; grep -E "forward|^a b|^b c|worst" gpurun_out/r2e_determinism.log
