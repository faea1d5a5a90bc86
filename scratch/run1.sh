python -m pytest tests -m gpu -x -q -k "up_synthesis or modconv or upfirdn or networks or resample" 2>&1 | tail -15
