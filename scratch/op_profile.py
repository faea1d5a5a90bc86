"""aten-level profile of one benchmark step: which framework ops (outside libsbg_hip.so) carry the launch-bound tail, and who calls them"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

dev = torch.device('cuda', 0)
wl = bench.workload('sg2ada')
eng = bench.build_engine(dev, 1, 0, wl, batch=64, batch_gpu=32)
real = torch.rand(64, 3, 256, 256, device=dev) * 2 - 1
for _ in range(2):
    eng.train_iteration(real, None)
eng.batch_idx = 1           # a step without Dreg
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=False) as prof:
    eng.train_iteration(real, None)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=6)
rows = []
for e in ka:
    dt = getattr(e, 'self_device_time_total', None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt > 0:
        rows.append((dt, e.count, e.key, [s for s in e.stack if 'style-big-gan_amd' in s or 'bench' in s][:3]))
rows.sort(key=lambda r: -r[0])
tot = sum(r[0] for r in rows)
print(f'total device us {tot:.0f}')
agg = collections.defaultdict(lambda: [0.0, 0])
for dt, cnt, key, stack in rows:
    agg[key][0] += dt; agg[key][1] += cnt
print('--- by op')
for k, (dt, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f'{dt:10.0f} us {cnt:6d}  {k[:90]}')
print('--- by op + caller (ops under 30 us average only)')
n = 0
for dt, cnt, key, stack in rows:
    if dt / max(cnt, 1) < 30:
        print(f'{dt:9.0f} us {cnt:5d}  {key[:50]:50s} <- {" | ".join(s.split("/")[-1][:70] for s in stack)}')
        n += 1
        if n > 70:
            break
