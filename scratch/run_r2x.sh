#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python scratch/op_profile3.py sg2ada > gpurun_out/r2x_ops.log 2>&1; tail -62 gpurun_out/r2x_ops.log
