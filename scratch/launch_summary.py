"""per-(kind, dims) summary of a bench.py --launch-log file: python scratch/launch_summary.py FILE STEPS [TOP]"""
import json, collections, sys
rec = [json.loads(l) for l in open(sys.argv[1])]
steps = int(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for r in rec:
    a = agg[(r['kind'], tuple(r['dims']))]; a[0] += 1; a[1] += r['ms']; a[2] += r['bytes']; a[3] += r['flops']
print('total ms/step', sum(v[1] for v in agg.values()) / steps)
fam = collections.defaultdict(lambda: [0.0, 0.0])
for (kind, dims), (n, ms, by, fl) in agg.items():
    f = fam[(kind, dims[6] // 1000000 if kind in ('conv_igemm', 'conv_wgrad') else 0)]; f[0] += ms; f[1] += fl
for k, (ms, fl) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f'  family {k}: {ms / steps:7.3f} ms/step  {fl / ms / 1e9 if ms else 0:7.0f} TF')
for (kind, dims), (n, ms, by, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f'{kind:14s} {str(dims):52s} n={n:4d} {ms / steps:7.3f} ms/step avg {ms / n * 1e3:7.1f} us  {by / ms / 1e6:7.0f} GB/s {fl / ms / 1e9:7.0f} TF')
