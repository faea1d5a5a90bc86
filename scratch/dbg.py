import torch, sys
sys.path.insert(0,'.')
import style_big_gan_amd
from style_big_gan_amd.torch_utils.ops import bias_act
from oracle import ops as O
torch.manual_seed(1)
dev='cuda'
dtype=torch.bfloat16
for shape in [(2,5,4,4),(3,16,6,5)]:
  for clamp,gain in [(None,None),(0.5,2.0)]:
    x0=torch.randn(shape); b0=torch.randn(shape[1])
    xq,bq=x0.to(dtype).float(),b0.to(dtype).float()
    xr=xq.clone().requires_grad_(True)
    yr=O.bias_act(xr,bq,act='selu',gain=gain,clamp=clamp)
    xg=xq.to(dev,dtype).requires_grad_(True)
    yg=bias_act.bias_act(xg,bq.to(dev,dtype),act='selu',gain=gain,clamp=clamp)
    dy=torch.randn(shape).to(dtype).float()
    gr=torch.autograd.grad((yr*dy).sum(),xr)[0]
    gg=torch.autograd.grad((yg*dy.to(dev,dtype)).sum(),xg)[0].float().cpu()
    d=(gg-gr).abs(); i=d.argmax()
    print(shape,clamp,gain,'maxdiff',d.max().item(),'at x+b',(xq+bq.view(1,-1,1,1)).flatten()[i].item(),'yr',yr.flatten()[i].item(),'yg',yg.float().cpu().flatten()[i].item(),'gr',gr.flatten()[i].item(),'gg',gg.flatten()[i].item(),'dy',dy.flatten()[i].item())
