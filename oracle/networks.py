"""CPU restatement of the StyleGAN2 generator / discriminator forward passes (test infrastructure, see oracle/__init__.py).

Functional style: every function takes the reference's ``state_dict`` (name -> tensor; names exactly as produced by
``train_parts/generators.py`` / ``discriminators.py`` / ``stylegan2ada/training/networks.py``) and a small config dict,
and returns plain differentiable fp32 PyTorch results -- so one code path serves forward parity, gradient parity
(make the state_dict tensors leaves with requires_grad) and the CPU baseline timing.

Reference lines followed: FullyConnectedLayer train_parts/generators.py:105-134; Conv2dLayer :139-185; MappingNetwork
:190-267; SynthesisLayer :273-329; ToRGBLayer :334-348; SynthesisBlock :354-458; SynthesisNetwork :464-519;
DiscriminatorBlock discriminators.py:211-302; MinibatchStdLayer :307-328; DiscriminatorEpilogue :332-389;
Discriminator :403-466.
"""
import numpy as np
import torch

from . import ops as O

RESAMPLE = [1, 3, 3, 1]


def default_cfg(**kw):
    cfg = dict(z_dim=512, c_dim=0, w_dim=512, img_resolution=256, img_channels=3, channel_base=32768, channel_max=512,
               mapping_layers=8, g_architecture='skip', d_architecture='resnet', conv_clamp=None, mbstd_group_size=4,
               mbstd_num_channels=1, w_avg_beta=0.995, lr_multiplier=0.01,
               g_attentions=(), d_attentions=())       # block resolutions with the non-local attention hook (generators.py:443-445, discriminators.py:297-299)
    cfg.update(kw)
    return cfg


def _channels(cfg, res):
    return min(cfg['channel_base'] // res, cfg['channel_max'])


def _f():
    return O.setup_filter(RESAMPLE)


# ---------------------------------------------------------------------------------------------------------------- layers

def fully_connected(sd, prefix, x, activation='linear', lr_multiplier=1.0):
    w = sd[prefix + '.weight']
    w = w * (lr_multiplier / np.sqrt(w.shape[1]))
    b = sd.get(prefix + '.bias')
    if b is not None and lr_multiplier != 1:
        b = b * lr_multiplier
    if activation == 'linear' and b is not None:
        return torch.addmm(b.unsqueeze(0), x, w.t())
    return O.bias_act(x.matmul(w.t()), b, act=activation)


def conv2d_layer(sd, prefix, x, activation='linear', up=1, down=1, conv_clamp=None, gain=1.0):
    w = sd[prefix + '.weight']
    k = w.shape[2]
    w = w * (1 / np.sqrt(w.shape[1] * k * k))
    b = sd.get(prefix + '.bias')
    x = O.conv2d_resample(x, w, f=_f(), up=up, down=down, padding=k // 2, flip_weight=(up == 1))
    act_gain = O.ACTIVATIONS[activation][2] * gain
    clamp = conv_clamp * gain if conv_clamp is not None else None
    return O.bias_act(x, b, act=activation, gain=act_gain, clamp=clamp)


def normalize_2nd_moment(x, dim=1, eps=1e-8):
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()


def mapping(sd, prefix, z, c, cfg, num_ws, z_dim=None, num_layers=None):
    z_dim = cfg['z_dim'] if z_dim is None else z_dim
    num_layers = cfg['mapping_layers'] if num_layers is None else num_layers
    x = None
    if z_dim > 0:
        x = normalize_2nd_moment(z.to(torch.float32))
    if cfg['c_dim'] > 0:
        y = normalize_2nd_moment(fully_connected(sd, prefix + '.embed', c.to(torch.float32)))
        x = torch.cat([x, y], dim=1) if x is not None else y
    for i in range(num_layers):
        x = fully_connected(sd, f'{prefix}.fc{i}', x, activation='lrelu', lr_multiplier=cfg['lr_multiplier'])
    if num_ws is not None:
        x = x.unsqueeze(1).repeat([1, num_ws, 1])
    return x


def synthesis_layer(sd, prefix, x, w, up, conv_clamp, noise_mode='const', noise=None, gain=1.0, fused_modconv=False):
    styles = fully_connected(sd, prefix + '.affine', w)
    res = x.shape[2] * up
    nz = None
    if noise is not None:
        nz = noise * sd[prefix + '.noise_strength']
    elif noise_mode == 'const':
        nz = sd[prefix + '.noise_const'] * sd[prefix + '.noise_strength']
    x = O.modulated_conv2d(x, sd[prefix + '.weight'], styles, noise=nz, up=up, padding=1, resample_filter=_f(),
                           flip_weight=(up == 1), fused_modconv=fused_modconv)
    act_gain = O.ACTIVATIONS['lrelu'][2] * gain
    clamp = conv_clamp * gain if conv_clamp is not None else None
    assert x.shape[2] == res
    return O.bias_act(x, sd[prefix + '.bias'], act='lrelu', gain=act_gain, clamp=clamp)


def torgb_layer(sd, prefix, x, w, conv_clamp, fused_modconv=False):
    wt = sd[prefix + '.weight']
    styles = fully_connected(sd, prefix + '.affine', w) * (1 / np.sqrt(wt.shape[1] * wt.shape[2] ** 2))
    x = O.modulated_conv2d(x, wt, styles, demodulate=False, fused_modconv=fused_modconv)
    return O.bias_act(x, sd[prefix + '.bias'], clamp=conv_clamp)


def synthesis_num_ws(cfg):
    log2 = int(np.log2(cfg['img_resolution']))
    n = 0
    for res in [2 ** i for i in range(2, log2 + 1)]:
        n += 1 if res == 4 else 2
    return n + 1        # + toRGB of the last block


def synthesis(sd, prefix, ws, cfg, noise_mode='const', noises=None, fused_modconv=False, sn_updates=None):
    """noises: optional dict layer-prefix -> [N, 1, R, R] tensor replacing the random draw of noise_mode='random'"""
    log2 = int(np.log2(cfg['img_resolution']))
    arch = cfg['g_architecture']
    clamp = cfg['conv_clamp']
    ws = ws.to(torch.float32)
    x = img = None
    w_idx = 0
    for res in [2 ** i for i in range(2, log2 + 1)]:
        bp = f'{prefix}.b{res}'
        is_last = res == cfg['img_resolution']
        nconv = 1 if res == 4 else 2
        cur = ws[:, w_idx:w_idx + nconv + 1]
        w_idx += nconv
        wi = 0
        nz = (lambda name: None if noises is None else noises.get(name))
        if res == 4:
            x = sd[bp + '.const'].unsqueeze(0).repeat([ws.shape[0], 1, 1, 1])
            x = synthesis_layer(sd, bp + '.conv1', x, cur[:, wi], 1, clamp, noise_mode, nz(bp + '.conv1'), fused_modconv=fused_modconv); wi += 1
        elif arch == 'resnet':
            y = conv2d_layer(sd, bp + '.skip', x, up=2, gain=np.sqrt(0.5))
            x = synthesis_layer(sd, bp + '.conv0', x, cur[:, wi], 2, clamp, noise_mode, nz(bp + '.conv0'), fused_modconv=fused_modconv); wi += 1
            x = synthesis_layer(sd, bp + '.conv1', x, cur[:, wi], 1, clamp, noise_mode, nz(bp + '.conv1'), gain=np.sqrt(0.5), fused_modconv=fused_modconv); wi += 1
            x = y + x
        else:
            x = synthesis_layer(sd, bp + '.conv0', x, cur[:, wi], 2, clamp, noise_mode, nz(bp + '.conv0'), fused_modconv=fused_modconv); wi += 1
            x = synthesis_layer(sd, bp + '.conv1', x, cur[:, wi], 1, clamp, noise_mode, nz(bp + '.conv1'), fused_modconv=fused_modconv); wi += 1
        if res in cfg.get('g_attentions', ()):          # end of the block, in fp32, spectral-norm power iteration advanced once (training mode)
            from . import biggan as OB
            x = OB.attention(sd, bp + '.attention', x, True, sn_updates if sn_updates is not None else {})
        if img is not None:
            img = O.upsample2d(img, _f())
        if is_last or arch == 'skip':
            y = torgb_layer(sd, bp + '.torgb', x, cur[:, wi], clamp, fused_modconv=fused_modconv)
            img = img + y if img is not None else y
    return img


def generator(sd, z, c, cfg, noise_mode='const', noises=None, fused_modconv=False, sn_updates=None):
    ws = mapping(sd, 'mapping', z, c, cfg, num_ws=synthesis_num_ws(cfg))
    return synthesis(sd, 'synthesis', ws, cfg, noise_mode=noise_mode, noises=noises, fused_modconv=fused_modconv, sn_updates=sn_updates)


def minibatch_std(x, group_size, num_channels=1):
    N, C, H, W = x.shape
    G = min(group_size, N) if group_size is not None else N
    F, c = num_channels, C // num_channels
    y = x.reshape(G, -1, F, c, H, W)
    y = y - y.mean(dim=0)
    y = (y.square().mean(dim=0) + 1e-8).sqrt()
    y = y.mean(dim=[2, 3, 4]).reshape(-1, F, 1, 1).repeat(G, 1, H, W)
    return torch.cat([x, y], dim=1)


def discriminator(sd, img, c, cfg, sn_updates=None):
    log2 = int(np.log2(cfg['img_resolution']))
    arch = cfg['d_architecture']
    clamp = cfg['conv_clamp']
    x = None
    img = img.to(torch.float32)
    for res in [2 ** i for i in range(log2, 2, -1)]:
        bp = f'b{res}'
        if x is None or arch == 'skip':
            y = conv2d_layer(sd, bp + '.fromrgb', img, activation='lrelu', conv_clamp=clamp)
            x = x + y if x is not None else y
            img = O.downsample2d(img, _f()) if arch == 'skip' else None
        if arch == 'resnet':
            y = conv2d_layer(sd, bp + '.skip', x, down=2, gain=np.sqrt(0.5))
            x = conv2d_layer(sd, bp + '.conv0', x, activation='lrelu', conv_clamp=clamp)
            x = conv2d_layer(sd, bp + '.conv1', x, activation='lrelu', down=2, conv_clamp=clamp, gain=np.sqrt(0.5))
            x = y + x
        else:
            x = conv2d_layer(sd, bp + '.conv0', x, activation='lrelu', conv_clamp=clamp)
            x = conv2d_layer(sd, bp + '.conv1', x, activation='lrelu', down=2, conv_clamp=clamp)
        if res in cfg.get('d_attentions', ()):
            from . import biggan as OB
            x = OB.attention(sd, bp + '.attention', x, True, sn_updates if sn_updates is not None else {})
    # epilogue
    if arch == 'skip':
        x = x + conv2d_layer(sd, 'b4.fromrgb', img, activation='lrelu')
    if cfg['mbstd_num_channels'] > 0:
        x = minibatch_std(x, cfg['mbstd_group_size'], cfg['mbstd_num_channels'])
    x = conv2d_layer(sd, 'b4.conv', x, activation='lrelu', conv_clamp=clamp)
    x = fully_connected(sd, 'b4.fc', x.flatten(1), activation='lrelu')
    x = fully_connected(sd, 'b4.out', x)
    if cfg['c_dim'] > 0:
        cmap_dim = sd['b4.out.weight'].shape[0]
        cmap = mapping(sd, 'mapping', None, c, cfg, num_ws=None, z_dim=0)
        x = (x * cmap).sum(dim=1, keepdim=True) * (1 / np.sqrt(cmap_dim))
    return x


# ---------------------------------------------------------------------------------------------------------------- training step

def softplus_losses():
    import torch.nn.functional as F
    g = lambda fake: F.softplus(-fake).mean()
    d = lambda real, fake: F.softplus(-real).mean() + F.softplus(fake).mean()
    return g, d


def gd_step_grads(g_sd, d_sd, cfg, z_g, z_d, real, r1_gamma=None, noise_mode='const', c=None):
    """Gradients of one Gmain + Dmain (+ R1) pass with softplus losses (train_parts/losses_base.py:50-109,
    regularizations.py:41-56).  Returns (loss_G, grads_G, loss_D, grads_D, grads_R1 or None); the grads are dicts keyed by
    state_dict name.  `c`: labels [N, c_dim] (default: empty / zeros)."""
    g_loss_fn, d_loss_fn = softplus_losses()
    if c is None:
        c = torch.zeros([z_g.shape[0], cfg['c_dim']])
    frozen = lambda k: ('noise_const' in k or 'resample' in k or 'w_avg' in k)

    def leaves(sd):
        return {k: v.detach().clone().requires_grad_(v.is_floating_point() and not frozen(k)) for k, v in sd.items()}

    def grads_of(loss, leaf, retain=False):
        names = [k for k, v in leaf.items() if v.requires_grad]
        gs = torch.autograd.grad(loss, [leaf[k] for k in names], allow_unused=True, retain_graph=retain)
        return {k: (g if g is not None else torch.zeros_like(leaf[k])) for k, g in zip(names, gs)}

    # Gmain: G forward -> D forward -> gradient into G only
    g_leaf = leaves(g_sd)
    d_const = {k: v.detach() for k, v in d_sd.items()}
    fake = generator(g_leaf, z_g, c, cfg, noise_mode=noise_mode)
    loss_g = g_loss_fn(discriminator(d_const, fake, c, cfg))
    grads_g = grads_of(loss_g, g_leaf)
    # Dmain: G forward without graph, D on fake and real
    d_leaf = leaves(d_sd)
    with torch.no_grad():
        fake = generator({k: v.detach() for k, v in g_sd.items()}, z_d, c, cfg, noise_mode=noise_mode)
    real_in = real.detach().clone().requires_grad_(r1_gamma is not None)
    real_logits = discriminator(d_leaf, real_in, c, cfg)
    loss_d = d_loss_fn(real_logits, discriminator(d_leaf, fake, c, cfg))
    grads_d = grads_of(loss_d, d_leaf, retain=r1_gamma is not None)
    grads_r1 = None
    if r1_gamma is not None:    # R1: double backward through D
        r1 = torch.autograd.grad(real_logits.sum(), real_in, create_graph=True)[0]
        pen = (r1.square().sum([1, 2, 3]) * (r1_gamma / 2)).mean()
        grads_r1 = grads_of(pen, d_leaf)
    return loss_g.detach(), grads_g, loss_d.detach(), grads_d, grads_r1
