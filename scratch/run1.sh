python -m pytest tests -m gpu -x -q -k "conv" 2>&1 | tail -2
for v in 3 2; do echo "== NSTAGE=$v"; SBG_WGRAD_NSTAGE=$v timeout -k 10 120 python scratch/kbench.py wgrad2 2>&1 | grep "conv_wgrad"; done
