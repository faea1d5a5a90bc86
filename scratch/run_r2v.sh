#!/bin/bash
for abl in 0 16 32 2 48; do SBG_K64_ABL=$abl python scratch/abl.py 2>&1 | grep TF; done
