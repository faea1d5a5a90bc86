"""BigGAN building blocks over the HIP op layer.

Mirror of the reference's ``biggan/layers.py`` API used by the registered models: ``SN`` / ``SNConv2d`` / ``SNLinear`` /
``SNEmbedding`` (:60-138), ``Attention`` (:144-169), ``ccbn`` (:278-330), ``bn`` (:333-366), ``GBlock`` (:375-409),
``DBlock`` (:412-457), ``identity`` (:54) with the same constructor arguments, buffer names (``u0``, ``sv0``, ``stored_mean``,
``stored_var``) and forward semantics, so state_dicts interchange.  Underneath:

* spectral norm: one fused power-iteration kernel pair (``sbg_sn_power_iteration``) gives v, u', sigma; the gradient
  d sigma / d W = outer(u', v) is an autograd Function, the weight is divided by sigma as in the reference (:99);
* convolutions are the implicit-GEMM MFMA kernels (``conv2d_gradfix``), ReLU is ``bias_act``, nearest up-sampling and 2x2
  average pooling are ``upfirdn2d`` with 2x2 box filters;
* attention: ``softmax(theta^T phi) g`` is one exact-fp32 matrix-core kernel (``sbg_attention_fwd``) -- the [HW, HW/4] map
  never reaches HBM; its backward re-derives the map with library batched GEMMs (differentiable again, so R1 works);
* batch norm: statistics by the ``sbg_dot_hw`` reductions (sum, sum of squares), normalise + class-conditional gain/bias in
  one ``sbg_scale_shift_nc`` pass; ``cross_replica=True`` all-reduces [sum, sum^2, count] over RCCL (differentiable) and
  uses the reference's synchronized formula (sync_batchnorm/batchnorm.py:147-158) -- the reference's DataParallel-only
  mechanism is not reproduced.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import Parameter as P

from .. import _lib
from ..torch_utils.ops import bias_act, conv2d_gradfix, modulate, upfirdn2d


class identity(nn.Module):
    def forward(self, input):
        return input


class ReLU(nn.Module):
    """out-of-place ReLU on the fused bias_act kernel (gain 1)"""

    def forward(self, x):
        if x.device.type != "cuda":
            return F.relu(x)
        return bias_act.bias_act(x, act="relu", gain=1)


def nearest_upsample2x(x):
    """F.interpolate(scale_factor=2) (nearest) as a zero-insert + 2x2 box FIR"""
    f = torch.ones([2, 2], dtype=torch.float32, device=x.device)
    return upfirdn2d.upfirdn2d(x, f, up=2, padding=[1, 0, 1, 0])


def avg_pool2x(x):
    """nn.AvgPool2d(2) as a 2x2 box FIR with decimation"""
    f = torch.full([2, 2], 0.25, dtype=torch.float32, device=x.device)
    return upfirdn2d.upfirdn2d(x, f, down=2)


# ---------------------------------------------------------------------------------------------------------------- spectral norm

class _SpectralSigma(torch.autograd.Function):
    """(W_mat [rows, cols], u [1, rows]) -> (sigma, u_new, v) by one power iteration; d sigma / d W = outer(u_new, v)"""

    @staticmethod
    def forward(ctx, W_mat, u, eps):
        lib = _lib.load()
        _lib.require_cuda(W_mat, "spectral norm")
        Wc = W_mat.detach().to(torch.float32).contiguous()
        rows, cols = Wc.shape
        uc = u.detach().reshape(rows).to(torch.float32).contiguous()
        v = torch.empty([cols], dtype=torch.float32, device=Wc.device)
        u_new = torch.empty([rows], dtype=torch.float32, device=Wc.device)
        sigma = torch.empty([1], dtype=torch.float32, device=Wc.device)
        ws = torch.empty([lib.sbg_sn_workspace(rows, cols) // 4], dtype=torch.float32, device=Wc.device)
        _lib.check(lib.sbg_sn_power_iteration(_lib.ptr(Wc), _lib.ptr(uc), _lib.ptr(v), _lib.ptr(u_new), _lib.ptr(sigma), _lib.ptr(ws),
                                              rows, cols, float(eps), _lib.stream_ptr(Wc.device)), "sbg_sn_power_iteration")
        ctx.save_for_backward(u_new, v)
        ctx.mark_non_differentiable(u_new, v)
        return sigma.reshape([]), u_new.reshape(1, rows), v.reshape(1, cols)

    @staticmethod
    def backward(ctx, dsigma, _du, _dv):
        u_new, v = ctx.saved_tensors
        return dsigma * torch.outer(u_new, v), None, None


class SN(object):
    """spectral-norm mixin: one singular vector, one power iteration per forward (reference defaults)"""

    def __init__(self, num_svs, num_itrs, num_outputs, transpose=False, eps=1e-12):
        assert num_svs == 1 and num_itrs == 1, "only num_svs = num_itrs = 1 (the configurations the reference ships) is implemented"
        self.num_itrs, self.num_svs, self.transpose, self.eps = num_itrs, num_svs, transpose, eps
        for i in range(self.num_svs):
            self.register_buffer('u%d' % i, torch.randn(1, num_outputs))
            self.register_buffer('sv%d' % i, torch.ones(1))

    @property
    def u(self):
        return [getattr(self, 'u%d' % i) for i in range(self.num_svs)]

    @property
    def sv(self):
        return [getattr(self, 'sv%d' % i) for i in range(self.num_svs)]

    def W_(self):
        W_mat = self.weight.view(self.weight.size(0), -1)
        if self.transpose:
            W_mat = W_mat.t()
        sigma, u_new, _ = _SpectralSigma.apply(W_mat, self.u0, self.eps)
        if self.training:
            with torch.no_grad():
                self.u0.copy_(u_new)
                self.sv0.copy_(sigma.reshape(1))
        return self.weight / sigma


class SNConv2d(nn.Conv2d, SN):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 num_svs=1, num_itrs=1, eps=1e-12):
        nn.Conv2d.__init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        SN.__init__(self, num_svs, num_itrs, out_channels, eps=eps)

    def forward(self, x):
        return conv2d_gradfix.conv2d(x, self.W_().to(x.dtype), self.bias, self.stride, self.padding, self.dilation, self.groups)


class SNLinear(nn.Linear, SN):
    def __init__(self, in_features, out_features, bias=True, num_svs=1, num_itrs=1, eps=1e-12):
        nn.Linear.__init__(self, in_features, out_features, bias)
        SN.__init__(self, num_svs, num_itrs, out_features, eps=eps)

    def forward(self, x):
        return F.linear(x, self.W_(), self.bias)


class SNEmbedding(nn.Embedding, SN):
    def __init__(self, num_embeddings, embedding_dim, padding_idx=None, max_norm=None, norm_type=2, scale_grad_by_freq=False,
                 sparse=False, _weight=None, num_svs=1, num_itrs=1, eps=1e-12):
        nn.Embedding.__init__(self, num_embeddings, embedding_dim, padding_idx, max_norm, norm_type, scale_grad_by_freq, sparse, _weight)
        SN.__init__(self, num_svs, num_itrs, num_embeddings, eps=eps)

    def forward(self, x):
        return F.embedding(x, self.W_())


# ---------------------------------------------------------------------------------------------------------------- attention

def _attention_reference(theta, phi, g):
    """softmax(theta phi^T) g with library batched GEMMs: [N,Q,D], [N,M,D], [N,M,DV] -> [N,Q,DV] (differentiable)"""
    beta = F.softmax(torch.bmm(theta, phi.transpose(1, 2)), -1)
    return torch.bmm(beta, g)


class _AttentionCore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, theta, phi, g):
        lib = _lib.load()
        n, q, d = theta.shape
        m, dv = g.shape[1], g.shape[2]
        t32, p32, g32 = [t.to(torch.float32).contiguous() for t in (theta, phi, g)]
        out = torch.empty([n, q, dv], dtype=torch.float32, device=theta.device)
        _lib.check(lib.sbg_attention_fwd(_lib.ptr(t32), _lib.ptr(p32), _lib.ptr(g32), _lib.ptr(out), n, q, m, d, dv,
                                         _lib.stream_ptr(theta.device)), "sbg_attention_fwd")
        ctx.save_for_backward(theta, phi, g)
        return out.to(theta.dtype)

    @staticmethod
    def backward(ctx, dout):
        theta, phi, g = ctx.saved_tensors
        if torch.is_grad_enabled():
            # a regulariser differentiates this gradient again (R1 through D's attention): compose it from differentiable library ops ON
            # THE SAVED TENSORS THEMSELVES, so that the second-order graph reaches theta / phi / g as well as dout
            ins = [t if t.requires_grad else t.detach().requires_grad_(True) for t in (theta, phi, g)]
            with torch.enable_grad():
                out = _attention_reference(*ins)
                grads = torch.autograd.grad(out, ins, dout, create_graph=True)
        else:
            lib = _lib.load()
            n, q, d = theta.shape
            m, dv = g.shape[1], g.shape[2]
            if lib.sbg_attention_bwd_supported(q, m, d, dv):
                # first order: the recompute-softmax backward kernels (two passes, no [N, Q, M] map in HBM)
                t32, p32, g32, do32 = [t.to(torch.float32).contiguous() for t in (theta, phi, g, dout)]
                grads = [torch.empty_like(t) for t in (t32, p32, g32)]
                ws = torch.empty([lib.sbg_attention_bwd_workspace(n, q)], dtype=torch.uint8, device=theta.device)
                _lib.check(lib.sbg_attention_bwd(_lib.ptr(t32), _lib.ptr(p32), _lib.ptr(g32), _lib.ptr(do32), _lib.ptr(grads[0]), _lib.ptr(grads[1]),
                                                 _lib.ptr(grads[2]), _lib.ptr(ws), n, q, m, d, dv, _lib.stream_ptr(theta.device)), "sbg_attention_bwd")
                grads = [gr.to(t.dtype) for gr, t in zip(grads, (theta, phi, g))]
            else:
                with torch.enable_grad():
                    ins = [t.detach().requires_grad_(True) for t in (theta, phi, g)]
                    out = _attention_reference(*ins)
                grads = torch.autograd.grad(out, ins, dout)
        return tuple(gr if need else None for gr, need in zip(grads, ctx.needs_input_grad))


def attention_core(theta, phi, g):
    """theta [N, Q, D], phi [N, M, D], g [N, M, DV] -> softmax(theta phi^T, -1) g"""
    lib = _lib.load()
    if theta.device.type == "cuda" and lib.sbg_attention_supported(theta.shape[1], phi.shape[1], theta.shape[2], g.shape[2]):
        return _AttentionCore.apply(theta, phi, g)
    return _attention_reference(theta, phi, g)


class Attention(nn.Module):
    def __init__(self, ch, which_conv=SNConv2d, name='attention'):
        super().__init__()
        self.ch = ch
        self.which_conv = which_conv
        self.theta = self.which_conv(self.ch, self.ch // 8, kernel_size=1, padding=0, bias=False)
        self.phi = self.which_conv(self.ch, self.ch // 8, kernel_size=1, padding=0, bias=False)
        self.g = self.which_conv(self.ch, self.ch // 2, kernel_size=1, padding=0, bias=False)
        self.o = self.which_conv(self.ch // 2, self.ch, kernel_size=1, padding=0, bias=False)
        self.gamma = P(torch.tensor(0.), requires_grad=True)

    def forward(self, x, y=None):
        n, _, h, w = x.shape
        theta = self.theta(x)
        phi = F.max_pool2d(self.phi(x), [2, 2])
        g = F.max_pool2d(self.g(x), [2, 2])
        # [N, C', H, W] -> [N, pixels, C'] (row-major pixels, the order of the reference's .view)
        theta = theta.reshape(n, self.ch // 8, h * w).transpose(1, 2)
        phi = phi.reshape(n, self.ch // 8, h * w // 4).transpose(1, 2)
        g = g.reshape(n, self.ch // 2, h * w // 4).transpose(1, 2)
        o = attention_core(theta, phi, g)                                  # [N, HW, C/2]
        o = self.o(o.transpose(1, 2).reshape(n, self.ch // 2, h, w))
        return self.gamma * o + x


# ---------------------------------------------------------------------------------------------------------------- batch norm

class _AllReduceSum(torch.autograd.Function):
    """differentiable all-reduce(SUM) over the default process group (gradient = all-reduce(SUM) of the gradients)"""

    @staticmethod
    def forward(ctx, t):
        out = t.clone()
        torch.distributed.all_reduce(out)
        return out

    @staticmethod
    def backward(ctx, g):
        out = g.clone()
        torch.distributed.all_reduce(out)
        return out


def batch_stats(x, cross_replica=False):
    """per-channel (mean, biased var, unbiased var, count) over (N, H, W) [and over ranks]; fp32, differentiable"""
    n, c, h, w = x.shape
    if x.device.type == "cuda":
        m1, m2 = modulate.moments_hw(x)      # one pass over x for both moments (sbg_moments_hw)
        s1, s2 = m1.sum(0), m2.sum(0)
    else:   # plumbing path for CPU-only unit tests of the host logic
        xf = x.float()
        s1, s2 = xf.sum([0, 2, 3]), xf.square().sum([0, 2, 3])
    cnt = torch.full([1], float(n * h * w), dtype=torch.float32, device=x.device)
    if cross_replica and torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        packed = _AllReduceSum.apply(torch.cat([s1, s2, cnt]))
        s1, s2, cnt = packed[:c], packed[c:2 * c], packed[2 * c:]
    mean = s1 / cnt
    sumvar = s2 - s1 * mean
    return mean, sumvar / cnt, sumvar / (cnt - 1).clamp(min=1), cnt


def normalize(x, mean, var, gain, bias, eps):
    """(x - mean) * rsqrt(var + eps) * gain + bias with per-sample gain / bias [N, C] (or [1, C]); one streaming pass"""
    n, c = x.shape[0], x.shape[1]
    scale = torch.rsqrt(var + eps).reshape(1, c) * (gain.reshape(-1, c) if gain is not None else 1.0)
    shift = (bias.reshape(-1, c) if bias is not None else 0.0) - mean.reshape(1, c) * scale
    scale, shift = scale.expand(n, c), (shift.expand(n, c) if torch.is_tensor(shift) else torch.zeros_like(scale).expand(n, c))
    if x.device.type == "cuda":
        return modulate.scale_shift_nc(x, scale, shift)
    return x * scale.reshape(n, c, 1, 1).to(x.dtype) + shift.reshape(n, c, 1, 1).to(x.dtype)


def _moments_nc(x):
    """per-sample, per-channel (sum x, sum x^2) over (H, W) in fp32: [N, C] each, differentiable"""
    if x.device.type == "cuda":
        return modulate.moments_hw(x)
    xf = x.float()
    return xf.sum([2, 3]), xf.square().sum([2, 3])


def _scale_shift(x, scale, shift):
    """x * scale[n, c] + shift[n, c] in one streaming pass"""
    n, c = x.shape[0], x.shape[1]
    scale, shift = scale.expand(n, c), shift.expand(n, c)
    if x.device.type == "cuda":
        return modulate.scale_shift_nc(x, scale, shift)
    return x * scale.reshape(n, c, 1, 1).to(x.dtype) + shift.reshape(n, c, 1, 1).to(x.dtype)


def instance_norm(mod, x, gain, bias):
    """F.instance_norm(x, stored_mean, stored_var, None, None, training, 0.1, eps) * gain + bias (reference :317-319): statistics per sample
    and channel over (H, W) while training -- the running buffers receive the batch mean of the per-instance mean and UNBIASED variance, which
    is what torch's instance_norm does through its [1, N*C, H, W] batch-norm view --, the stored statistics otherwise."""
    n, c, h, w = x.shape
    if mod.training:
        s1, s2 = _moments_nc(x)
        hw = float(h * w)
        mean = s1 / hw
        var = (s2 / hw - mean.square()).clamp(min=0)
        with torch.no_grad():
            mod.stored_mean.mul_(0.9).add_(mean.detach().mean(0) * 0.1)
            mod.stored_var.mul_(0.9).add_((var.detach() * (hw / max(hw - 1, 1))).mean(0) * 0.1)
    else:
        mean, var = mod.stored_mean.reshape(1, c), mod.stored_var.reshape(1, c)
    scale = torch.rsqrt(var + mod.eps) * gain.reshape(-1, c)
    return _scale_shift(x, scale, bias.reshape(-1, c) - mean * scale)


def group_norm_groups(norm_style, channels):
    """number of groups a norm_style string asks for (reference `groupnorm`, :257-268): 'gn_ch_<k>' = k channels per group, 'gn_grp_<g>' = g
    groups, plain 'gn' = 16 groups"""
    if 'ch' in norm_style:
        return max(int(channels) // int(norm_style.split('_')[-1]), 1)
    if 'grp' in norm_style:
        return int(norm_style.split('_')[-1])
    return 16


def group_norm(x, groups, gain, bias, eps=1e-5):
    """F.group_norm(x, groups) * gain + bias: statistics per sample over (C / groups, H, W), biased variance, eps = F.group_norm's default
    (the reference does not pass its own, :268).  The reference's 'gn' branch reads a mistyped attribute (`self.normstyle`, :321) and cannot
    run; this is the behaviour it spells out."""
    n, c, h, w = x.shape
    if c % groups:
        raise ValueError(f"group norm: {c} channels are not divisible into {groups} groups")
    s1, s2 = _moments_nc(x)
    cnt = float((c // groups) * h * w)
    g1 = s1.reshape(n, groups, -1).sum(-1, keepdim=True) / cnt
    g2 = s2.reshape(n, groups, -1).sum(-1, keepdim=True) / cnt
    mean = g1.expand(n, groups, c // groups).reshape(n, c)
    var = (g2 - g1.square()).clamp(min=0).expand(n, groups, c // groups).reshape(n, c)
    scale = torch.rsqrt(var + eps) * gain.reshape(-1, c)
    return _scale_shift(x, scale, bias.reshape(-1, c) - mean * scale)


class myBN(nn.Module):
    """the reference's hand-written batch norm with standing statistics (:212-253): normalises with mean-of-squares minus squared-mean (BIASED
    variance), keeps running averages of the SAME biased variance -- unlike F.batch_norm, which stores the unbiased one -- or, with
    `accumulate_standing`, plain sums + a counter that evaluation divides by."""

    def __init__(self, num_channels, eps=1e-5, momentum=0.1):
        super().__init__()
        self.momentum, self.eps = momentum, eps
        self.register_buffer('stored_mean', torch.zeros(num_channels))
        self.register_buffer('stored_var', torch.ones(num_channels))
        self.register_buffer('accumulation_counter', torch.zeros(1))
        self.accumulate_standing = False
        self.cross_replica = False

    def reset_stats(self):
        self.stored_mean.zero_(); self.stored_var.zero_(); self.accumulation_counter.zero_()

    def forward(self, x, gain, bias):
        c = x.shape[1]
        if self.training:
            mean, var, _, _ = batch_stats(x, False)
            with torch.no_grad():
                if self.accumulate_standing:
                    self.stored_mean.add_(mean.detach()); self.stored_var.add_(var.detach()); self.accumulation_counter.add_(1.0)
                else:
                    self.stored_mean.mul_(1 - self.momentum).add_(mean.detach() * self.momentum)
                    self.stored_var.mul_(1 - self.momentum).add_(var.detach() * self.momentum)
        else:
            mean, var = self.stored_mean, self.stored_var
            if self.accumulate_standing:
                mean, var = mean / self.accumulation_counter, var / self.accumulation_counter
        return normalize(x, mean, var, gain.reshape(-1, c), bias.reshape(-1, c), self.eps)


NORM_STYLES = ('bn', 'in', 'gn', 'nonorm')


def _check_norm_style(norm_style):
    if norm_style not in NORM_STYLES and not norm_style.startswith('gn_'):
        raise NotImplementedError(f"norm_style={norm_style!r}: one of {NORM_STYLES} (or 'gn_ch_<k>' / 'gn_grp_<g>')")


class ccbn(nn.Module):
    """class-conditional batch norm: gain = 1 + Emb/Linear(y), bias = Emb/Linear(y) (reference :275-327).  `mybn` = the hand-written batch norm
    with standing statistics (buffers under `bn.`, as in the reference); norm_style 'bn' | 'in' | 'gn' | 'nonorm'."""

    def __init__(self, output_size, input_size, which_linear, eps=1e-5, momentum=0.1, cross_replica=False, mybn=False, norm_style='bn'):
        super().__init__()
        self.output_size, self.input_size = output_size, input_size
        self.gain = which_linear(input_size, output_size)
        self.bias = which_linear(input_size, output_size)
        self.eps, self.momentum = eps, momentum
        self.cross_replica, self.mybn, self.norm_style = cross_replica, mybn, norm_style
        _check_norm_style(norm_style)
        if self.mybn and not self.cross_replica:
            self.bn = myBN(output_size, self.eps, self.momentum)
        elif self.cross_replica or norm_style in ('bn', 'in'):
            # statistics live on this module ('stored_mean' / 'stored_var', the reference's buffer names for the default path)
            self.register_buffer('stored_mean', torch.zeros(output_size))
            self.register_buffer('stored_var', torch.ones(output_size))

    def forward(self, x, y):
        n = y.size(0)
        gain = (1 + self.gain(y)).view(n, -1)
        bias = self.bias(y).view(n, -1)
        if self.cross_replica:
            return _bn_forward(self, x, gain, bias, momentum=self.momentum)
        if self.mybn:
            return self.bn(x, gain, bias)
        if self.norm_style == 'bn':
            return _bn_forward(self, x, gain, bias, momentum=0.1)     # F.batch_norm(..., 0.1, eps) in the reference (:315-316)
        if self.norm_style == 'in':
            return instance_norm(self, x, gain, bias)
        if self.norm_style.startswith('gn'):
            return group_norm(x, group_norm_groups(self.norm_style, x.shape[1]), gain, bias)
        return _scale_shift(x, gain, bias)                            # 'nonorm'

    def extra_repr(self):
        return f'out: {self.output_size}, in: {self.input_size}, cross_replica={self.cross_replica}'


class bn(nn.Module):
    """plain batch norm with learned per-channel gain / bias (reference :331-364)"""

    def __init__(self, output_size, eps=1e-5, momentum=0.1, cross_replica=False, mybn=False):
        super().__init__()
        self.output_size = output_size
        self.gain = P(torch.ones(output_size), requires_grad=True)
        self.bias = P(torch.zeros(output_size), requires_grad=True)
        self.eps, self.momentum = eps, momentum
        self.cross_replica, self.mybn = cross_replica, mybn
        if self.mybn and not self.cross_replica:
            self.bn = myBN(output_size, self.eps, self.momentum)
        else:
            self.register_buffer('stored_mean', torch.zeros(output_size))
            self.register_buffer('stored_var', torch.ones(output_size))

    def forward(self, x, y=None):
        if self.mybn and not self.cross_replica:
            return self.bn(x, self.gain.view(1, -1), self.bias.view(1, -1))
        return _bn_forward(self, x, self.gain.view(1, -1), self.bias.view(1, -1), momentum=self.momentum)


def _bn_forward(mod, x, gain, bias, momentum):
    if mod.training:
        mean, var, unbiased, _ = batch_stats(x, mod.cross_replica)
        with torch.no_grad():
            mod.stored_mean.mul_(1 - momentum).add_(mean.detach() * momentum)
            mod.stored_var.mul_(1 - momentum).add_(unbiased.detach() * momentum)
    else:
        mean, var = mod.stored_mean, mod.stored_var
    return normalize(x, mean, var, gain, bias, mod.eps)


# ---------------------------------------------------------------------------------------------------------------- residual blocks

class GBlock(nn.Module):
    def __init__(self, in_channels, out_channels, which_conv=nn.Conv2d, which_bn=bn, activation=None, upsample=None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.which_conv, self.which_bn = which_conv, which_bn
        self.activation = activation
        self.upsample = upsample
        self.conv1 = self.which_conv(self.in_channels, self.out_channels)
        self.conv2 = self.which_conv(self.out_channels, self.out_channels)
        self.learnable_sc = in_channels != out_channels or upsample
        if self.learnable_sc:
            self.conv_sc = self.which_conv(in_channels, out_channels, kernel_size=1, padding=0)
        self.bn1 = self.which_bn(in_channels)
        self.bn2 = self.which_bn(out_channels)

    def forward(self, x, y):
        """residual generator block (reference :375-412): bn -> relu -> [2x nearest] -> conv -> bn -> relu -> conv, plus a 1x1 shortcut on the
        (up-sampled) input whenever the shape changes"""
        up = self.upsample if self.upsample else (lambda t: t)
        h = self.conv1(up(self.activation(self.bn1(x, y))))
        h = self.conv2(self.activation(self.bn2(h, y)))
        sc = up(x)
        return h + (self.conv_sc(sc) if self.learnable_sc else sc)


class DBlock(nn.Module):
    def __init__(self, in_channels, out_channels, which_conv=SNConv2d, wide=True, preactivation=False, activation=None, downsample=None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.hidden_channels = self.out_channels if wide else self.in_channels
        self.which_conv = which_conv
        self.preactivation = preactivation
        self.activation = activation
        self.downsample = downsample
        self.conv1 = self.which_conv(self.in_channels, self.hidden_channels)
        self.conv2 = self.which_conv(self.hidden_channels, self.out_channels)
        self.learnable_sc = True if (in_channels != out_channels) or downsample else False
        if self.learnable_sc:
            self.conv_sc = self.which_conv(in_channels, out_channels, kernel_size=1, padding=0)

    def shortcut(self, x):
        """1x1 convolution and 2x average pooling of the block input; pre-activation blocks convolve first, the first block pools first (:440-451)"""
        steps = [self.conv_sc if self.learnable_sc else None, self.downsample if self.downsample else None]
        if not self.preactivation:
            steps.reverse()
        for f in steps:
            if f is not None:
                x = f(x)
        return x

    def forward(self, x):
        h = self.activation(x) if self.preactivation else x     # out-of-place ReLU (it must not touch the shortcut's input)
        h = self.conv2(self.activation(self.conv1(h)))
        return (self.downsample(h) if self.downsample else h) + self.shortcut(x)
