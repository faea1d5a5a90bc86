#!/bin/bash
timeout -k 10 200 python scratch/up2_check.py 2>&1 | tail -12
SBG_CONV_NO_UP2=1 timeout -k 10 200 python scratch/up2_check.py 2>&1 | tail -4
