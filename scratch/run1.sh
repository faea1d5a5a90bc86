timeout -k 10 120 python scratch/k64_check.py 2>&1 | tail -1
python -m pytest tests -m gpu -x -q -k "conv or networks or modconv or resample" 2>&1 | tail -2
for v in 1 0; do echo "== NO_PHASES=$v"; if [ $v = 1 ]; then export SBG_CONV_NO_PHASES=1; else unset SBG_CONV_NO_PHASES; fi; timeout -k 10 120 python scratch/kbench.py conv 2>&1 | grep "convT"; done
