python -m pytest tests -m gpu -x -q -k "upfirdn or resample" 2>&1 | tail -2
timeout -k 10 120 python scratch/kbench.py fir 2>&1 | grep "fir"
