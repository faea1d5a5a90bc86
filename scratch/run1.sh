python -m pytest tests -m gpu -x -q -k "upfirdn or resample" 2>&1 | tail -12
for v in 0 1; do echo "== NO_MFMA=$v"; if [ $v = 1 ]; then export SBG_FIR_NO_MFMA=1; fi; timeout -k 10 120 python scratch/kbench.py fir 2>&1 | grep "fir"; done
