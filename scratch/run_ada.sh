#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_augment_gpu.py -x -q > gpurun_out/aug_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/aug_tests.log
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --ada 0.5 --kernel-breakdown > gpurun_out/ada_bench.json 2> gpurun_out/ada_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/ada_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'])
PY
