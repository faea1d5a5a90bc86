"""which lines of the package launch the framework's small kernels: one warm step under torch.profiler (with_stack), device kernels of non-library ops
grouped by the innermost package frame that issued them"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import style_big_gan_amd  # noqa: F401
from style_big_gan_amd import _lib
_lib.load()
dev = torch.device('cuda:0')
wl = bench.workload('sg2ada')
eng = bench.build_engine(dev, 1, 0, wl, batch=64, batch_gpu=32)
real = torch.rand(64, 3, 256, 256, device=dev) * 2 - 1
for _ in range(3):
    eng.train_iteration(real, None)
eng.batch_idx = 1            # Gmain + Dmain only
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    eng.train_iteration(real, None)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
        continue
    where = next((f for f in (ev.stack or []) if 'style-big-gan_amd' in f or 'parallel.py' in f), 'framework / autograd engine')
    where = where.split('style-big-gan_amd/')[-1][:90]
    k = (ev.name[:40], where)
    agg[k][0] += len(ev.kernels); agg[k][1] += sum(kk.duration for kk in ev.kernels)
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print(f'total device time of profiled ops: {tot / 1e3:.1f} ms')
for (name, where), (cnt, us) in rows[:70]:
    if us / max(cnt, 1) > 60:      # big kernels are not the subject here
        continue
    print(f'{us / 1e3:7.3f} ms {cnt:5d} x {us / max(cnt, 1):6.1f} us  {name:40s} {where}')
