timeout -k 10 120 python scratch/k64_check.py 2>&1 | tail -1
python -m pytest tests -m gpu -x -q -k "conv or networks or modconv" 2>&1 | tail -2
for v in 0 1; do echo "== GATHER_LD=$v"; SBG_K64_GATHER_LD=$v timeout -k 10 120 python scratch/kbench.py conv 2>&1 | grep "s2\|convT"; done
