#!/bin/bash
set -o pipefail
python scratch/determinism.py > gpurun_out/r2h_determinism.log 2>&1; grep -E "forward|first run|NaN|^a2 b|^b c|worst" gpurun_out/r2h_determinism.log
timeout -k 10 300 python scratch/thin_check.py > gpurun_out/r2h_thin.log 2>&1; tail -24 gpurun_out/r2h_thin.log
SBG_CONV_NO_THIN=1 timeout -k 10 300 python scratch/thin_check.py 2>&1 | tail -7 > gpurun_out/r2h_thin_off.log; cat gpurun_out/r2h_thin_off.log
