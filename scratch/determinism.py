"""is one StepEngine iteration a pure function of (state, inputs, seed)?  engines b and c resume from the same snapshot and run the same iteration"""
import io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import style_big_gan_amd
from style_big_gan_amd.train_parts import trainers
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_engine_gpu import _sg2_kwargs
dev = torch.device('cuda', 0)
gk, dk = _sg2_kwargs()
kw = dict(gen_kwargs=gk, disc_kwargs=dk, loss_arch_kwargs=dict(style_mixing_prob=0), dis_regs=[("r1", dict(r1_gamma=0.1))], g_reg_interval=4, d_reg_interval=2, batch=8, batch_gpu=4, ema_kimg=0.05)
a = trainers.StepEngine(dev, seed=1, **kw)
gen = torch.Generator(device=dev).manual_seed(3)
for _ in range(3):
    a.train_iteration(torch.rand(8, 3, 32, 32, device=dev, generator=gen) * 2 - 1, None)
buf = io.BytesIO(); torch.save(a.state_dict(), buf)
engines = {'a': a}
for name in 'bc':
    buf.seek(0)
    e = trainers.StepEngine(dev, seed=99, **kw)
    e.load_state_dict(torch.load(buf, map_location=dev, weights_only=True))
    engines[name] = e
for name, e in engines.items():
    for ph in e.phases:
        st = ph.opt.state_dict()['state']
        print(name, ph.name, 'steps', sorted({float(v['step']) for v in st.values()}), 'lr', ph.opt.param_groups[0]['lr'], 'betas', ph.opt.param_groups[0]['betas'],
              {k: v for k, v in ph.opt.param_groups[0].items() if k not in ('params', 'lr', 'betas')})
# forward-only probes from identical states
zz = torch.randn(4, 32, device=dev, generator=gen)
outs = {}
for name, e in engines.items():
    with torch.no_grad():
        torch.manual_seed(7)
        ws = e.G.mapping(zz, torch.zeros(4, 0, device=dev), skip_w_avg_update=True)
        img_c = e.G.synthesis(ws, noise_mode='const')
        torch.manual_seed(7)
        img_r = e.G.synthesis(ws, noise_mode='random')
        lg = e.D(img_c, torch.zeros(4, 0, device=dev))
    outs[name] = (ws, img_c, img_r, lg)
for q in 'bc':
    print('forward a vs', q, [float((x - y).abs().max()) for x, y in zip(outs['a'], outs[q])])
real = torch.rand(8, 3, 32, 32, device=dev, generator=gen) * 2 - 1
z = torch.randn(len(a.phases) * 8, 32, device=dev, generator=gen)
def poison():
    """fill the caching allocator's free blocks with NaN: a kernel that reads memory it (or its producer) never wrote then shows up"""
    blocks = [torch.full([n], float('nan'), device=dev) for n in (1 << 28, 1 << 27, 1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16) for _ in range(3)]
    blocks += [torch.full([n], float('nan'), device=dev, dtype=torch.bfloat16) for n in (1 << 28, 1 << 26, 1 << 24, 1 << 22, 1 << 20) for _ in range(3)]
    del blocks
grads = {}
for name, e in engines.items():
    if name == 'c':
        poison()
    torch.manual_seed(1234)
    e.train_iteration(real, None, all_gen_z=z)
    grads[name] = {k: v.grad.clone() for k, v in list(e.G.named_parameters()) + [('D.' + k, v) for k, v in e.D.named_parameters()] if v.grad is not None}
def cmp(x, y):
    worst = ('', 0.0)
    for (k, va), (_, vb) in zip(x.state_dict().items(), y.state_dict().items()):
        d = float((va.float() - vb.float()).abs().max())
        if d > worst[1]:
            worst = (k, d)
    return worst
print('first run: a vs b  G', cmp(engines['a'].G, engines['b'].G), 'D', cmp(engines['a'].D, engines['b'].D))
print('NaN in c after poisoned iteration:', any(bool(torch.isnan(v).any()) for v in list(engines['c'].G.state_dict().values()) + list(engines['c'].D.state_dict().values())))
# a again, from the restored snapshot: does reloading remove the difference?
buf.seek(0)
a.load_state_dict(torch.load(buf, map_location=dev, weights_only=True))
torch.manual_seed(1234)
a.train_iteration(real, None, all_gen_z=z)
grads['a2'] = {k: v.grad.clone() for k, v in list(a.G.named_parameters()) + [('D.' + k, v) for k, v in a.D.named_parameters()] if v.grad is not None}
engines['a2'] = a
for p, q in (('b', 'c'), ('a2', 'b')):
    print(p, q, 'G', cmp(engines[p].G, engines[q].G), 'D', cmp(engines[p].D, engines[q].D))
    gw = max(((k, float((grads[p][k] - grads[q][k]).abs().max() / (grads[q][k].abs().max() + 1e-30))) for k in grads[p]), key=lambda t: t[1])
    print('   last-phase grads: worst relative difference', gw)
