#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_augment_gpu.py tests/test_ops_gpu.py -x -q -k "augment or upfirdn or grid or pipe or filter or training_step" > gpurun_out/aug_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/aug_tests.log
ADA_ARGS="--ada 0.5" bash scratch/prof_ada.sh
python - <<'PY'
import json
d=json.loads(open('gpurun_out/ada_prof_bench.json').read().strip().splitlines()[-1])
print('under rocprof', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])
PY
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --ada 0.5 > gpurun_out/ada_bench.json 2> gpurun_out/ada_bench.err; echo "ada bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/ada_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'])
PY
