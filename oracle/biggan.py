"""CPU restatement of the BigGAN layers and models (test infrastructure, see oracle/__init__.py).

Functional, state_dict-driven like oracle/networks.py.  Training-mode buffer updates (spectral-norm ``u0`` / ``sv0``, batch-norm
``stored_mean`` / ``stored_var``) are written into the ``updates`` dict the caller passes, never in place.

Reference lines followed: power_iteration biggan/layers.py:28-50; SN.W_ :87-99; SNConv2d/SNLinear/SNEmbedding :103-138;
Attention :144-169; ccbn :278-325; bn :333-366; GBlock :375-409; DBlock :412-457; BigGAnGenerator
train_parts/generators.py:756-937 (G_shared=False, hier=False, G_param='SN', norm_style='bn'); BigGanDiscriminator
train_parts/discriminators.py:733-842; synchronized statistics biggan/sync_batchnorm/batchnorm.py:147-158.
"""
import numpy as np
import torch
import torch.nn.functional as F


def power_iteration(W, u, eps=1e-12):
    """one step, one singular vector: returns (sigma with grad through W, u_new, v)"""
    with torch.no_grad():
        v = F.normalize(torch.matmul(u, W), eps=eps)
        u_new = F.normalize(torch.matmul(v, W.t()), eps=eps)
    sigma = torch.squeeze(torch.matmul(torch.matmul(v, W.t()), u_new.t()))
    return sigma, u_new, v


def sn_weight(sd, prefix, training, updates, eps=1e-12):
    w = sd[prefix + '.weight']
    sigma, u_new, _ = power_iteration(w.reshape(w.shape[0], -1), sd[prefix + '.u0'], eps)
    if training and updates is not None:
        updates[prefix + '.u0'] = u_new.detach()
        updates[prefix + '.sv0'] = sigma.detach().reshape(1)
    return w / sigma


def sn_conv(sd, prefix, x, padding, training, updates):
    return F.conv2d(x, sn_weight(sd, prefix, training, updates), sd.get(prefix + '.bias'), 1, padding)


def sn_linear(sd, prefix, x, training, updates):
    return F.linear(x, sn_weight(sd, prefix, training, updates), sd.get(prefix + '.bias'))


def sn_embedding(sd, prefix, idx, training, updates):
    return F.embedding(idx, sn_weight(sd, prefix, training, updates))


def attention(sd, prefix, x, training, updates):
    n, ch, h, w = x.shape
    theta = sn_conv(sd, prefix + '.theta', x, 0, training, updates)
    phi = F.max_pool2d(sn_conv(sd, prefix + '.phi', x, 0, training, updates), [2, 2])
    g = F.max_pool2d(sn_conv(sd, prefix + '.g', x, 0, training, updates), [2, 2])
    theta = theta.view(-1, ch // 8, h * w)
    phi = phi.view(-1, ch // 8, h * w // 4)
    g = g.view(-1, ch // 2, h * w // 4)
    beta = F.softmax(torch.bmm(theta.transpose(1, 2), phi), -1)
    o = sn_conv(sd, prefix + '.o', torch.bmm(g, beta.transpose(1, 2)).view(-1, ch // 2, h, w), 0, training, updates)
    return sd[prefix + '.gamma'] * o + x


def _batch_norm(sd, prefix, x, training, updates, momentum, eps):
    if training:
        mean = x.mean([0, 2, 3])
        var = x.var([0, 2, 3], unbiased=False)
        if updates is not None:
            n = x.numel() / x.shape[1]
            updates[prefix + '.stored_mean'] = (1 - momentum) * sd[prefix + '.stored_mean'] + momentum * mean.detach()
            updates[prefix + '.stored_var'] = (1 - momentum) * sd[prefix + '.stored_var'] + momentum * var.detach() * n / (n - 1)
    else:
        mean, var = sd[prefix + '.stored_mean'], sd[prefix + '.stored_var']
    return (x - mean.view(1, -1, 1, 1)) * torch.rsqrt(var.view(1, -1, 1, 1) + eps)


def ccbn(sd, prefix, x, y_idx, training, updates, eps=1e-5):
    gain = (1 + F.embedding(y_idx, sd[prefix + '.gain.weight'])).view(y_idx.size(0), -1, 1, 1)
    bias = F.embedding(y_idx, sd[prefix + '.bias.weight']).view(y_idx.size(0), -1, 1, 1)
    return _batch_norm(sd, prefix, x, training, updates, 0.1, eps) * gain + bias


def bn(sd, prefix, x, training, updates, eps=1e-5, momentum=0.1):
    out = _batch_norm(sd, prefix, x, training, updates, momentum, eps)
    return out * sd[prefix + '.gain'].view(1, -1, 1, 1) + sd[prefix + '.bias'].view(1, -1, 1, 1)


def synchronized_stats(s1, s2, count, eps=1e-5):
    """mean / inv_std / unbiased var from all-reduced sums (sync_batchnorm/batchnorm.py:147-158)"""
    mean = s1 / count
    sumvar = s2 - s1 * mean
    return mean, torch.rsqrt(sumvar / count + eps), sumvar / (count - 1)


def gblock(sd, prefix, x, y_idx, training, updates, upsample=True):
    h = F.relu(ccbn(sd, prefix + '.bn1', x, y_idx, training, updates))
    if upsample:
        h = F.interpolate(h, scale_factor=2)
        x = F.interpolate(x, scale_factor=2)
    h = sn_conv(sd, prefix + '.conv1', h, 1, training, updates)
    h = F.relu(ccbn(sd, prefix + '.bn2', h, y_idx, training, updates))
    h = sn_conv(sd, prefix + '.conv2', h, 1, training, updates)
    if (prefix + '.conv_sc.weight') in sd:
        x = sn_conv(sd, prefix + '.conv_sc', x, 0, training, updates)
    return h + x


def dblock(sd, prefix, x, training, updates, preactivation, downsample):
    h = F.relu(x) if preactivation else x
    h = sn_conv(sd, prefix + '.conv1', h, 1, training, updates)
    h = sn_conv(sd, prefix + '.conv2', F.relu(h), 1, training, updates)
    if downsample:
        h = F.avg_pool2d(h, 2)
    sc = x
    has_sc = (prefix + '.conv_sc.weight') in sd
    if preactivation:
        if has_sc:
            sc = sn_conv(sd, prefix + '.conv_sc', sc, 0, training, updates)
        if downsample:
            sc = F.avg_pool2d(sc, 2)
    else:
        if downsample:
            sc = F.avg_pool2d(sc, 2)
        if has_sc:
            sc = sn_conv(sd, prefix + '.conv_sc', sc, 0, training, updates)
    return h + sc


def generator(sd, z, c, training=True, updates=None, bottom_width=4):
    y = torch.argmax(c, dim=1)
    h = sn_linear(sd, 'linear', z, training, updates)
    h = h.view(h.size(0), -1, bottom_width, bottom_width)
    i = 0
    while f'blocks.{i}.0.conv1.weight' in sd:
        h = gblock(sd, f'blocks.{i}.0', h, y, training, updates)
        if f'blocks.{i}.1.gamma' in sd:
            h = attention(sd, f'blocks.{i}.1', h, training, updates)
        i += 1
    h = F.relu(bn(sd, 'output_layer.0', h, training, updates))
    return torch.tanh(sn_conv(sd, 'output_layer.2', h, 1, training, updates))


def discriminator(sd, x, c, downsample_flags, training=True, updates=None):
    y = torch.argmax(c, dim=1)
    h = x
    for i, down in enumerate(downsample_flags):
        h = dblock(sd, f'blocks.{i}.0', h, training, updates, preactivation=(i > 0), downsample=down)
        if f'blocks.{i}.1.gamma' in sd:
            h = attention(sd, f'blocks.{i}.1', h, training, updates)
    h = torch.sum(F.relu(h), [2, 3])
    out = sn_linear(sd, 'linear', h, training, updates)
    return out + torch.sum(sn_embedding(sd, 'embed', y, training, updates) * h, 1, keepdim=True)


D_DOWNSAMPLE = {32: [True, True, False, False], 64: [True] * 4 + [False], 128: [True] * 5 + [False], 256: [True] * 6 + [False]}
