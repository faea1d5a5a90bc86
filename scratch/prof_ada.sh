#!/bin/bash
export TMPDIR=/tmp
OUT=/tmp/adaprof; rm -rf $OUT; mkdir -p $OUT gpurun_out
rocprofv3 --kernel-trace --stats -d $OUT -o ada --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --ada 0.5 > gpurun_out/ada_prof_bench.json 2> $OUT/log.txt
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
head -40 $f | cut -c1-260 > gpurun_out/ada_kernel_stats.txt
echo done
