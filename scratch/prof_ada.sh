#!/bin/bash
export TMPDIR=/tmp
OUT=/tmp/adaprof; rm -rf $OUT; mkdir -p $OUT gpurun_out
rocprofv3 --kernel-trace --stats -d $OUT -o ada --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline ${ADA_ARGS:---ada 0.5} > gpurun_out/ada_prof_bench.json 2> $OUT/log.txt
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/adaprof/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r['TotalDurationNs']) for r in rows)
with open('gpurun_out/ada_kernel_stats.txt', 'w') as o:
    o.write(f"total_ms\t{tot/1e6:.2f}\tkernels\t{sum(int(r['Calls']) for r in rows)}\n")
    for r in rows[:90]:
        o.write(f"{int(r['TotalDurationNs'])/1e6:.2f}\t{r['Calls']}\t{float(r['AverageNs'])/1e3:.1f}\t{r['Name'][:220]}\n")
PY
echo done
