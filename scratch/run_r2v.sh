#!/bin/bash
for abl in 0 4 1 2 8 5 12; do SBG_K64_ABL=$abl python scratch/abl.py 2>&1 | grep TF; done
