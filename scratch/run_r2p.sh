#!/bin/bash
python scratch/op_profile2.py big_gan > gpurun_out/r2p_ops_big_gan.log 2>&1; head -60 gpurun_out/r2p_ops_big_gan.log | tail -56
python scratch/op_profile2.py sg2attent > gpurun_out/r2p_ops_sg2attent.log 2>&1; echo; head -50 gpurun_out/r2p_ops_sg2attent.log | tail -46
