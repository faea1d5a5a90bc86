"""who launches the large framework copies / elementwise kernels of one headline step: python scratch/op_profile3.py [workload]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
name = sys.argv[1] if len(sys.argv) > 1 else 'sg2ada'
dev = torch.device('cuda', 0)
wl = bench.workload(name)
eng = bench.build_engine(dev, 1, 0, wl, batch=wl['batch'], batch_gpu=wl['batch_gpu'])
real = torch.rand(wl['batch'], 3, wl['res'], wl['res'], device=dev) * 2 - 1
c = torch.nn.functional.one_hot(torch.arange(wl['batch'], device=dev) % 10, 10).float() if wl['c_dim'] else None
for _ in range(2):
    eng.train_iteration(real, c)
eng.batch_idx = 0
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    eng.train_iteration(real, c)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0, None])
for e in prof.events():
    dt = getattr(e, 'self_device_time_total', 0) or 0
    if dt <= 0 or not e.name.startswith('aten::'):
        continue
    stack = [s for s in (e.stack or []) if 'style-big-gan_amd' in s or 'style_big_gan_amd' in s or 'bench.py' in s]
    key = (e.name, str(e.input_shapes)[:90], stack[0][-70:] if stack else '')
    a = agg[key]; a[0] += dt; a[1] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
print('aten device us total', sum(v[0] for v in agg.values()))
for (n, sh, st), (dt, cnt, _) in rows[:60]:
    print(f'{dt:9.0f} us {cnt:5d} {n:22s} {sh:90s} {st}')
