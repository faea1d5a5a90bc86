#!/bin/bash
# Run on the GPU box from the repo root:  bash profiles/collect.sh <tag>
# Three rocprofv3 passes over the same bench command (kernel trace + stats; FETCH_SIZE; WRITE_SIZE -- the two TCC counters do
# not fit one pass, and gpurun refuses --pmc together with API tracing), then profiles/summarize.py condenses them.
set -e
set -o pipefail
TAG=${1:-r01}
export TMPDIR=/tmp
CMD="python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline"
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o $TAG --output-format csv -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
echo "stats pass done"
# the PMC passes run under a time limit: a counter-collection failure of the profiler must not hang the box (the summary then lacks traffic)
timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o $TAG --output-format csv -- $CMD > /dev/null 2> $OUT/fetch.log || echo "FETCH_SIZE pass failed: $(tail -2 $OUT/fetch.log)"
echo "fetch pass done"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o $TAG --output-format csv -- $CMD > /dev/null 2> $OUT/write.log || echo "WRITE_SIZE pass failed: $(tail -2 $OUT/write.log)"
echo "write pass done"
python3 profiles/summarize.py $OUT $TAG
# keep only the condensed files in gpurun_out (the raw traces exceed what gpurun merges back)
mkdir -p gpurun_out/profiles_out && cp profiles/${TAG}_* gpurun_out/profiles_out/ && rm -rf $OUT
