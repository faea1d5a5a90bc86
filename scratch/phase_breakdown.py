"""per-phase device time of the headline step: library kernels by kind (launch log) and everything else (phase wall on the device minus the log)"""
import sys, collections
sys.path.insert(0, '.')
import torch
import bench
from style_big_gan_amd import _lib
dev = torch.device('cuda:0')
wl = bench.workload('sg2ada', None, None)
eng = bench.build_engine(dev, 1, 0, wl, batch=wl['batch'], batch_gpu=wl['batch_gpu'])
real_u8 = torch.randint(0, 256, [wl['batch'], 3, wl['res'], wl['res']], device=dev, dtype=torch.uint8)
def step():
    eng.train_iteration(real_u8.to(torch.float32) / 127.5 - 1, None)
for _ in range(5): step()
torch.cuda.synchronize()
stats = collections.defaultdict(lambda: [0.0, collections.Counter(), collections.Counter(), 0])
inner = eng.loss.accumulate_gradients
def wrapped(phase, **kw):
    torch.cuda.synchronize(); _lib.prof_enable(True); _lib.prof_fetch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); inner(phase=phase, **kw); e1.record(); torch.cuda.synchronize(); _lib.prof_enable(False)
    st = stats[phase]; st[0] += e0.elapsed_time(e1); st[3] += 1
    for r in _lib.prof_fetch():
        st[1][r['kind']] += r['ms']; st[2][r['kind']] += 1
eng.loss.accumulate_gradients = wrapped
eng.batch_idx = 0
for _ in range(4): step()
for ph, (ms, kinds, counts, runs) in stats.items():
    lib = sum(kinds.values())
    print(f'{ph}: {ms / runs:7.2f} ms per run ({runs} runs), library kernels {lib / runs:7.2f} ms, other {(ms - lib) / runs:6.2f} ms')
    for k, v in kinds.most_common(): print(f'      {k:14s} {v / runs:7.2f} ms  {counts[k] / runs:6.1f} launches')
