import sys, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import style_big_gan_amd
from golden_util import Golden, max_rel
import test_networks_gpu as T
import torch.nn.functional as F
dev=torch.device('cuda:0')
def poison():
    t=[torch.full((64<<20,), float('nan'), device=dev) for _ in range(8)]
    del t
g=Golden('networks_skip_resnet')
G,D=T.build(g,dev)
z,z2,c,real=g.t('z').to(dev),g.t('z2').to(dev),g.t('c').to(dev),g.t('real').to(dev)
G.requires_grad_(False); D.requires_grad_(True)
res=[]
for it in range(3):
    poison()
    with torch.no_grad():
        fake=G.synthesis(G.mapping(z2,c,skip_w_avg_update=True),noise_mode='const')
    real_in=real.clone().requires_grad_(True)
    real_logits=D(real_in,c)
    loss_d=F.softplus(-real_logits).mean()+F.softplus(D(fake,c)).mean()
    for p in D.parameters(): p.grad=None
    loss_d.backward()
    grads={n:p.grad.clone() for n,p in D.named_parameters()}
    errs=[(max_rel(p.grad,g.t('gradD/'+n)),n) for n,p in D.named_parameters()]
    print(it,'Dmain worst',sorted(errs)[-3:], 'nan' , any(torch.isnan(v).any().item() for v in grads.values()))
    res.append(grads)
for n in res[0]:
    if not torch.equal(res[0][n],res[1][n]) or not torch.equal(res[1][n],res[2][n]):
        print('nondeterministic',n,(res[0][n]-res[1][n]).abs().max().item())
