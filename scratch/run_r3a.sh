#!/bin/bash
# kernel-trace stats pass only (what runs outside the library's kernels after the merged rounds)
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_r3a
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o r3a --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/stats.log
f=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
cp $f gpurun_out/r3a_kernel_stats.csv
rm -rf $OUT/stats
cat $OUT/bench.json | cut -c1-300
