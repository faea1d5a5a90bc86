#!/bin/bash
# the ragged-column split of the matrix-core FIR: parity tests, then the bench with the split off / on on one box
python -m pytest tests/test_ops_gpu.py tests/test_reference_vectors_gpu.py tests/test_networks_gpu.py -m gpu -x -q > gpurun_out/r3i_test.log 2>&1; tail -2 gpurun_out/r3i_test.log
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export SBG_FIR_NO_EDGE=1; else unset SBG_FIR_NO_EDGE; fi
  python bench.py --no-cpu-baseline --kernel-breakdown > gpurun_out/r3i_bench_$v.json 2> gpurun_out/r3i_bench_$v.err
  echo "NO_EDGE=$v $(python -c "import json;d=json.load(open('gpurun_out/r3i_bench_$v.json'));print(d['value'],d['ms_per_step'],d['kernel_ms_per_step']['upfirdn2d'])")"
  grep "upfirdn2d " gpurun_out/r3i_bench_$v.err | head -4
done
