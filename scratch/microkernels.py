"""Where do the step's framework micro-kernels (fp32 adds, fills, small sums, copies, addmm) come from?  One training step of the headline workload
under torch.profiler with python stacks; device time of every non-library kernel grouped by (aten op, innermost frame inside this package).
    python scratch/microkernels.py > gpurun_out/microkernels.txt"""
import sys, collections
sys.path.insert(0, '.')
import torch
import bench
from torch.profiler import profile, ProfilerActivity

dev = torch.device('cuda:0')
wl = bench.workload('sg2ada', None, None)
eng = bench.build_engine(dev, 1, 0, wl, batch=wl['batch'], batch_gpu=wl['batch_gpu'])
res = wl['res']
real_u8 = torch.randint(0, 256, [wl['batch'], 3, res, res], device=dev, dtype=torch.uint8)
def step():
    real = real_u8.to(torch.float32) / 127.5 - 1
    eng.train_iteration(real, None)
for _ in range(5): step()
torch.cuda.synchronize()
eng.batch_idx = 0
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(4): step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, set()])
for ev in prof.events():
    dt = getattr(ev, 'self_device_time_total', 0) or 0
    if dt <= 0 or not ev.name.startswith('aten::'): continue
    frame = next((f for f in (ev.stack or []) if 'style_big_gan_amd' in f or 'style-big-gan_amd' in f), None) or next(iter(ev.stack or ['?']), '?')
    if frame == '?' or 'style' not in frame:       # backward nodes have no python stack: name the autograd node from the op's thread-local parent instead
        p = ev.cpu_parent
        while p is not None and not (p.name.startswith('autograd::') or 'Backward' in p.name): p = p.cpu_parent
        frame = 'backward of ' + p.name if p is not None else frame
    k = (ev.name, frame.split('/')[-1][:110])
    a = agg[k]; a[0] += 1; a[1] += dt; a[2].add(str(ev.input_shapes)[:70])
tot = sum(a[1] for a in agg.values())
print(f'aten device time: {tot / 4e3:.3f} ms/step over {sum(a[0] for a in agg.values()) / 4:.0f} ops/step')
for (name, frame), (n, dt, shapes) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:90]:
    print(f'{dt / 4e3:7.3f} ms/step {n / 4:6.1f}/step  {name:28s} {frame:112s} {sorted(shapes)[:2]}')
