"""merged rounds vs separate rounds (tests/test_engine_gpu.py::test_rounds_in_one_pass_equal_separate_rounds, fp32 networks): per-tensor error
ratios, with the style bank on and off"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import style_big_gan_amd
from style_big_gan_amd.train_parts import trainers, generators
from test_engine_gpu import _sg2_kwargs
dev = torch.device('cuda:0')
real = torch.rand(16, 3, 32, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(21)) * 2 - 1
allres = {}
for bank in (False, True):
    generators.style_bank_enabled = bank
    gk, dk = _sg2_kwargs(res=32, nfp=0)
    kw = dict(gen_kwargs=gk, disc_kwargs=dk, loss_arch_kwargs=dict(style_mixing_prob=0), dis_regs=[("r1", dict(r1_gamma=0.1))], g_reg_interval=4,
              d_reg_interval=4, batch=16, batch_gpu=8, ema_kimg=0.05)
    out = []
    for merge in (True, False):
        trainers.merge_rounds = merge
        eng = trainers.StepEngine(dev, seed=5, **kw)
        for m in eng.G.synthesis.modules():
            if hasattr(m, 'use_noise'): m.use_noise = False
        z = torch.randn(len(eng.phases) * 16, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(9))
        grads = {}
        for ph in eng.phases:
            def step(ph=ph, inner=ph.opt.step):
                grads[ph.name] = [(n, p.grad.detach().clone()) for n, p in ph.module.named_parameters() if p.grad is not None]
                return inner()
            ph.opt.step = step
        eng.train_iteration(real, None, all_gen_z=z)
        out.append(grads); eng.close()
    ga, gb = out
    allres[bank] = out
    rows = []
    for name in gb:
        for (n, a), (_, b) in zip(ga[name], gb[name]):
            rows.append((float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12), name, n, tuple(b.shape), float(b.abs().max())))
    rows.sort(reverse=True)
    print(f'style bank {bank}:')
    for r in [r for r in rows if r[4] > 1e-5][:8]: print('   %.3e  %-6s %-40s %s max %.3e' % r)

for idx, label in ((0, 'merged'), (1, 'separate')):
    ga, gb = allres[True][idx], allres[False][idx]
    rows = []
    for name in gb:
        for (n, a), (_, b) in zip(ga[name], gb[name]):
            rows.append((float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12), name, n, tuple(b.shape), float(b.abs().max())))
    rows.sort(reverse=True)
    print(f'{label}: bank on vs bank off')
    for r in [r for r in rows if r[4] > 1e-5][:6]: print('   %.3e  %-6s %-40s %s max %.3e' % r)
