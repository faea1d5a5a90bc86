#!/bin/bash
set -o pipefail
timeout -k 10 300 python scratch/thin_wgrad_check.py > gpurun_out/r2k_thin_wgrad.log 2>&1; tail -14 gpurun_out/r2k_thin_wgrad.log
SBG_WGRAD_NO_THIN=1 timeout -k 10 300 python scratch/thin_wgrad_check.py 2>&1 | tail -3
