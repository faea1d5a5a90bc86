python -m pytest tests -m gpu -x -q -k "conv" 2>&1 | tail -2
for v in 64 128; do echo "== BCA=$v"; SBG_WGRAD_BCA=$v timeout -k 10 120 python scratch/kbench.py wgrad wgrad2 2>&1 | grep "conv_wgrad"; done
