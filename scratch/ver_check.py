import torch
p = torch.nn.Parameter(torch.randn(8, 8, device='cuda'))
for fused in (False, True):
    opt = torch.optim.Adam([p], lr=0.1, **(dict(fused=True) if fused else {}))
    p.grad = torch.randn_like(p)
    v0 = p._version; before = p.detach().clone()
    opt.step()
    print('fused', fused, 'version', v0, '->', p._version, 'changed', bool((p.detach() != before).any()))
