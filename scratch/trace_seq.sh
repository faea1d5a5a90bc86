#!/bin/bash
# ordered kernel sequence of ONE training step (compact) -> gpurun_out/seq.txt
set -e
export TMPDIR=/tmp
OUT=/tmp/seqprof; rm -rf $OUT; mkdir -p $OUT gpurun_out
rocprofv3 --kernel-trace -d $OUT -o seq --output-format csv -- python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline > gpurun_out/seq_bench.json 2> $OUT/log.txt
python3 - <<'PY'
import csv, glob, re
f = glob.glob('/tmp/seqprof/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'at::native::', '', n)
    n = re.sub(r'void ', '', n)
    m = re.match(r'(vectorized_elementwise_kernel|elementwise_kernel_manual_unroll|unrolled_elementwise_kernel|elementwise_kernel)<[^,]*,\s*(.*)', n)
    if m:
        n = 'ew:' + m.group(2)
    return n[:110]
# last third of the trace = the timed step (warmup 2 + 1 step)
t0 = int(rows[0]['Start_Timestamp']); 
n = len(rows)
with open('gpurun_out/seq.txt', 'w') as o:
    prev_end = None
    for r in rows[n * 2 // 3 - 200:]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = (s - prev_end) / 1e3 if prev_end else 0
        prev_end = e
        o.write(f"{(s - t0) / 1e6:10.3f} {(e - s) / 1e3:8.1f} {gap:6.1f} g{r.get('Grid_Size_X', r.get('Grid_Size','?'))} {short(r['Kernel_Name'])}\n")
print('rows', n)
PY
ls -la gpurun_out/seq.txt
