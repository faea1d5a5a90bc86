import sys, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import style_big_gan_amd
from style_big_gan_amd.torch_utils.ops import conv_bias_act, bias_act, conv2d_gradfix
dev='cuda'
torch.manual_seed(9)
def rel(a,b): return float((a.float()-b.float()).abs().max()/(b.float().abs().max()+1e-12))
for (stride, pad, act, clamp, gain) in [(1, 1, "lrelu", 0.8, None), (2, 0, "lrelu", None, 0.7), (1, 0, "linear", 0.5, None), (1, 1, "relu", None, None), (1,1,"lrelu",None,None)]:
    k = 1 if pad == 0 and stride == 1 else 3
    x = torch.randn(2, 24, 17, 17, device=dev).to(torch.bfloat16).requires_grad_(True)
    w = (torch.randn(40, 24, k, k, device=dev) / (24 * k * k) ** 0.5).to(torch.bfloat16).requires_grad_(True)
    b = torch.randn(40, device=dev).to(torch.bfloat16).requires_grad_(True)
    y_f = conv_bias_act.conv2d_bias_act(x, w, b, stride=stride, padding=pad, act=act, gain=gain, clamp=clamp)
    y_u = bias_act.bias_act(conv2d_gradfix.conv2d(x, w, stride=stride, padding=pad), b, act=act, gain=gain, clamp=clamp)
    dy = torch.randn_like(y_f)
    gf = torch.autograd.grad((y_f * dy).sum(), [x, w, b])
    gu = torch.autograd.grad((y_u * dy).sum(), [x, w, b])
    # fp32 reference
    xr=x.detach().float().requires_grad_(True); wr=w.detach().float().requires_grad_(True); br=b.detach().float().requires_grad_(True)
    from oracle import ops as O
    yr=O.bias_act(torch.nn.functional.conv2d(xr.cpu(),wr.cpu(),stride=stride,padding=pad), br.cpu(), act=act, gain=gain, clamp=clamp)
    print((stride,pad,act,clamp,gain),'y f/u',rel(y_f,y_u),'y f/ref',rel(y_f.cpu(),yr),'y u/ref',rel(y_u.cpu(),yr),'grads f/u',[round(rel(a,c),4) for a,c in zip(gf,gu)])
