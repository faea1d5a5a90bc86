#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/full_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/full_tests.log
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --kernel-breakdown > gpurun_out/full_bench.json 2> gpurun_out/full_bench_breakdown.log; echo "bench rc=$?"
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --ada 0.5 --kernel-breakdown > gpurun_out/ada_bench.json 2> gpurun_out/ada_bench.err; echo "ada bench rc=$?"
python - <<'PY'
import json
for f in ('gpurun_out/full_bench.json','gpurun_out/ada_bench.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['roofline']['achieved'], d['kernel_ms_per_step'])
PY
