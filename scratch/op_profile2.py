"""framework-op profile of one step of a secondary workload: python scratch/op_profile2.py big_gan|sg2attent"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
name = sys.argv[1]
dev = torch.device('cuda', 0)
wl = bench.workload(name)
eng = bench.build_engine(dev, 1, 0, wl, batch=wl['batch'], batch_gpu=wl['batch_gpu'])
real = torch.rand(wl['batch'], 3, wl['res'], wl['res'], device=dev) * 2 - 1
c = torch.nn.functional.one_hot(torch.arange(wl['batch'], device=dev) % 10, 10).float() if wl['c_dim'] else None
for _ in range(2):
    eng.train_iteration(real, c)
eng.batch_idx = 0
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    eng.train_iteration(real, c)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    dt = getattr(e, 'self_device_time_total', None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt > 0:
        rows.append((dt, e.count, e.key))
rows.sort(key=lambda r: -r[0])
print('total device us', sum(r[0] for r in rows))
for dt, cnt, key in rows[:70]:
    print(f'{dt:10.0f} us {cnt:6d}  {key[:110]}')
