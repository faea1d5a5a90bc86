import sys, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import style_big_gan_amd
from golden_util import Golden, max_rel
from style_big_gan_amd.torch_utils.ops import conv2d_gradfix
import test_networks_gpu as T
import torch.nn.functional as F
dev=torch.device('cuda:0')
for passes in [3,6]:
    conv2d_gradfix.fp32_mfma_passes = passes
    g=Golden('networks_skip_resnet')
    G,D=T.build(g,dev)
    z,z2,c,real=g.t('z').to(dev),g.t('z2').to(dev),g.t('c').to(dev),g.t('real').to(dev)
    G.requires_grad_(False); D.requires_grad_(True)
    with torch.no_grad():
        fake=G.synthesis(G.mapping(z2,c,skip_w_avg_update=True),noise_mode='const')
    real_in=real.clone().requires_grad_(True)
    real_logits=D(real_in,c)
    loss_d=F.softplus(-real_logits).mean()+F.softplus(D(fake,c)).mean()
    loss_d.backward(retain_graph=True)
    errs=[(max_rel(p.grad,g.t('gradD/'+n)),n) for n,p in D.named_parameters()]
    print(passes,'Dmain worst',sorted(errs)[-3:])
    for p in D.parameters(): p.grad=None
    r1=torch.autograd.grad(real_logits.sum(),real_in,create_graph=True)[0]
    pen=(r1.square().sum([1,2,3])*(0.5/2)).mean(); pen.backward()
    errs=[(max_rel(p.grad if p.grad is not None else torch.zeros_like(p),g.t('gradR1/'+n)),n) for n,p in D.named_parameters()]
    print(passes,'R1 worst',sorted(errs)[-4:], 'pen',float(pen),float(g.t('r1_penalty')))
