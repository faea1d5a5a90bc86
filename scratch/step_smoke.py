import sys, time, torch
sys.path.insert(0, '.')
import style_big_gan_amd
from style_big_gan_amd.train_parts import trainers
res = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cb = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8
nfp16 = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device('cuda:0')
gk = dict(z_dim=512, c_dim=0, w_dim=512, img_resolution=res, img_channels=3,
          mapping_kwargs=dict(num_layers=2), synthesis_kwargs=dict(channel_base=cb, num_fp16_res=nfp16, block_kwargs=dict(conv_clamp=256)))
dk = dict(c_dim=0, img_resolution=res, img_channels=3, architecture='orig', channel_base=cb, num_fp16_res=nfp16, conv_clamp=256,
          epilogue_kwargs=dict(mbstd_group_size=min(batch, 32)))
eng = trainers.StepEngine(dev, gen_kwargs=gk, disc_kwargs=dk, loss_arch_kwargs=dict(style_mixing_prob=0),
                          dis_regs=[('r1', dict(r1_gamma=0.01))], batch=batch, batch_gpu=batch, ema_kimg=0.5)
real = (torch.randint(0, 256, [batch, 3, res, res], device=dev).float() / 127.5 - 1)
for it in range(6):
    torch.cuda.synchronize(); t = time.time()
    eng.train_iteration(real, None)
    torch.cuda.synchronize(); print('iter', it, 'ms', (time.time() - t) * 1e3, flush=True)
g = sum(float(p.grad.abs().sum()) for p in eng.D.parameters())
print('D grad abs sum', g, 'finite params', all(torch.isfinite(p).all() for p in eng.G.parameters()))
