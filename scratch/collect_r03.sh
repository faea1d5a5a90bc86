#!/bin/bash
# round 3 evidence run: smoke, rocprofv3 passes (profiles/collect.sh <tag>), default bench (driver's command), secondary workloads
TAG=${1:-r03a}
mkdir -p gpurun_out
python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/${TAG}_smoke.log
bash profiles/collect.sh $TAG > gpurun_out/collect_$TAG.log 2>&1; echo "collect rc=$?"; tail -4 gpurun_out/collect_$TAG.log
timeout -k 10 500 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "default bench rc=$?"
cp gpurun_out/${TAG}_bench_default.json profiles/${TAG}_bench_default.json
: > gpurun_out/${TAG}_secondary.jsonl
for args in "--workload ffhq_sg2" "--workload big_gan" "--workload sg2attent" "--workload sg2attent --res 256" "--ada 0.5" "--num-fp16-res 0"; do
  timeout -k 10 400 python bench.py $args --no-cpu-baseline --no-secondary >> gpurun_out/${TAG}_secondary.jsonl 2>> gpurun_out/${TAG}_secondary.err || echo "FAILED: $args" >> gpurun_out/${TAG}_secondary.jsonl
  echo "done: $args"
done
cp gpurun_out/${TAG}_secondary.jsonl profiles/${TAG}_secondary_workloads.jsonl
mkdir -p gpurun_out/profiles_out && cp profiles/${TAG}_* gpurun_out/profiles_out/
python - $TAG <<'PY'
import json, sys
tag = sys.argv[1]
r = json.load(open(f'gpurun_out/{tag}_bench_default.json'))
print({k: r[k] for k in ('value', 'ms_per_step', 'ms_per_step_median', 'target_kernel', 'sbg_kernel_time_frac_of_step')}, r['roofline']['frac'], [s_['value'] for s_ in r.get('secondary', [])], r['cpu_baseline']['value'])
for l in open(f'gpurun_out/{tag}_secondary.jsonl'):
    try:
        d = json.loads(l); print(d['metric'], d['value'], d['ms_per_step'])
    except Exception: print(l.strip()[:100])
PY
