"""Generators (host-side modules over the HIP op layer).

Mirror of the registry surface and module tree of the reference's ``train_parts/generators.py``: the registry
``generators`` with ``@generators.add_to_registry("sg2_classic" | "cnn32_dcgan" | "cnn48_dcgan" | "big_gan")``, and for
StyleGAN2 the modules ``FullyConnectedLayer`` (:105), ``Conv2dLayer`` (:139), ``MappingNetwork`` (:190),
``SynthesisLayer`` (:273), ``ToRGBLayer`` (:334), ``SynthesisBlock`` (:354), ``SynthesisNetwork`` (:464) and
``Generator`` (:533) with identical constructor arguments, parameter / buffer names (state_dicts interchange with the
reference) and forward semantics.  What differs is underneath:

* every tensor op on the hot path is a hand-written gfx950 kernel (bias_act, upfirdn2d, implicit-GEMM convolutions,
  per-sample scaling) reached through ``torch_utils.ops``;
* the reduced-precision blocks (``use_fp16`` in the reference) run in ``LOW_PRECISION`` = bfloat16 by default --
  the matrix-core dtype of choice on MI355X -- and always channel-minor; ``set_low_precision(torch.float16)``
  restores the reference's dtype;
* ``modulated_conv2d`` never materialises per-sample weights: demodulation coefficients come from
  ``styles^2 @ sum_kk(w^2)^T`` (algebraically the reference's :71-76) and both ``fused_modconv`` settings scale
  activations around one shared-weight convolution (the reference's training path :79-88; its eval-only grouped
  convolution :90-100 computes the same function).
"""
import numpy as np
import torch

from .. import utils
from ..torch_utils import misc
from ..torch_utils.ops import bias_act, conv2d_gradfix, conv2d_resample, conv_bias_act, grouped_gemm, modconv, modulate, torgb, upfirdn2d

generators = utils.ClassRegistry()

LOW_PRECISION = torch.bfloat16
import os as _os
style_bank_enabled = _os.environ.get('SBG_STYLE_BANK', '1') != '0'   # the styles of all layers of a pass from one launch (SynthesisNetwork._style_bank)
premodulate = _os.environ.get('SBG_PREMODULATE', '1') != '0'     # inference passes: conv0's tail applies conv1's style modulation (SynthesisBlock.forward)


def set_low_precision(dtype):
    """dtype used by blocks constructed with use_fp16=True: torch.bfloat16 (default) or torch.float16"""
    global LOW_PRECISION
    assert dtype in (torch.bfloat16, torch.float16)
    LOW_PRECISION = dtype


def low_precision():
    return LOW_PRECISION


# ----------------------------------------------------------------------------------------------------------------
# StyleGAN2

@misc.profiled_function
def normalize_2nd_moment(x, dim=1, eps=1e-8):
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()


def demod_coefficients(weight, styles, act_dtype=None):
    """dcoefs[n, o] = rsqrt(sum_{i,kh,kw} (w[o,i,kh,kw] * s[n,i])^2 + 1e-8), as a [N, I] x [I, O] product (fp32).
    `act_dtype` (16-bit, first-order passes of the fused layers): the HIP path that shares the packed-weight pass (ops/modconv.py)."""
    if (act_dtype in (torch.bfloat16, torch.float16) and modconv.enabled and weight.dtype == torch.float32 and weight.device.type == 'cuda'
            and styles.device.type == 'cuda'):
        return modconv.demod_coefs(weight, styles, act_dtype)
    w2 = weight.to(torch.float32).square().sum(dim=[2, 3])            # [O, I]
    return (styles.to(torch.float32).square() @ w2.t() + 1e-8).rsqrt()   # [N, O]


@misc.profiled_function
def modulated_conv2d(
    x,                          # [N, Cin, H, W]
    weight,                     # [Cout, Cin, kh, kw]
    styles,                     # [N, Cin] modulation coefficients
    noise           = None,     # optional noise added to the output
    up              = 1,
    down            = 1,
    padding         = 0,        # w.r.t. the upsampled image
    resample_filter = None,     # from upfirdn2d.setup_filter()
    demodulate      = True,
    flip_weight     = True,     # False = convolution, True = correlation
    fused_modconv   = True,     # accepted for API parity; both settings run the same kernels (see module docstring)
):
    n = x.shape[0]
    cout, cin, kh, kw = weight.shape
    misc.assert_shape(x, [n, cin, None, None])
    misc.assert_shape(styles, [n, cin])

    if x.dtype == torch.float16 and demodulate:     # keep fp16 activations in range (reference :63-65); bf16 needs none
        weight = weight * (1 / np.sqrt(cin * kh * kw) / weight.norm(float("inf"), dim=[1, 2, 3], keepdim=True))
        styles = styles / styles.norm(float("inf"), dim=1, keepdim=True)

    dcoefs = demod_coefficients(weight, styles) if demodulate else None
    x = modulate.scale_nc(x, styles)
    x = conv2d_resample.conv2d_resample(x=x, w=weight.to(x.dtype), f=resample_filter, up=up, down=down,
                                        padding=padding, flip_weight=flip_weight)
    if demodulate:
        x = modulate.scale_nc(x, dcoefs, noise)
    elif noise is not None:
        x = x.add_(noise.to(x.dtype))
    return x


class FullyConnectedLayer(torch.nn.Module):
    def __init__(self,
        in_features,
        out_features,
        bias            = True,
        activation      = 'linear',
        lr_multiplier   = 1,
        bias_init       = 0,
    ):
        super().__init__()
        self.activation = activation
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = torch.nn.Parameter(torch.full([out_features], np.float32(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / np.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def forward(self, x):
        b = self.bias
        if b is not None:
            b = b.to(x.dtype)
            if self.bias_gain != 1:
                b = b * self.bias_gain
        w = self.weight.to(x.dtype)
        if self.activation == 'linear' and b is not None:
            return _scaled_linear(x, w, b, float(self.weight_gain))
        return bias_act.bias_act(_scaled_linear(x, w, None, float(self.weight_gain)), b, act=self.activation)


_zero_cache = {}


def _zero(like):
    """a one-element zero tensor per (device, dtype): the ignored `input` of addmm(..., beta=0)"""
    key = (like.device, like.dtype)
    z = _zero_cache.get(key)
    if z is None:
        z = _zero_cache[key] = torch.zeros([1], device=like.device, dtype=like.dtype)
    return z


class _ScaledLinear(torch.autograd.Function):
    """y = b + alpha * x @ W^T with the gain in the GEMM's alpha, forward and in both backward products (reference: `w = weight * weight_gain`
    then addmm / matmul, train_parts/generators.py:117-127 -- a pass over W per call, and autograd's backward of it is mm, mul, mm, mul).
    First-order calls run three kernels backward (two GEMMs, the bias sum); when a graph is being built (R1, path length) the backward is the
    differentiable composition of the same products."""

    @staticmethod
    def forward(ctx, x, w, b, alpha):
        ctx.save_for_backward(x, w)
        ctx.alpha = alpha
        ctx.has_bias = b is not None
        if b is not None:
            return torch.addmm(b.unsqueeze(0), x, w.t(), alpha=alpha)
        return torch.addmm(_zero(x), x, w.t(), beta=0, alpha=alpha)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        alpha = ctx.alpha
        dx = dw = db = None
        if torch.is_grad_enabled():
            if ctx.needs_input_grad[0]:
                dx = g.mm(w) * alpha
            if ctx.needs_input_grad[1]:
                dw = g.t().mm(x) * alpha
        else:
            if ctx.needs_input_grad[0]:
                dx = torch.addmm(_zero(g), g, w, beta=0, alpha=alpha)
            if ctx.needs_input_grad[1]:
                dw = torch.addmm(_zero(g), g.t(), x, beta=0, alpha=alpha)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = g.sum(0)
        return dx, dw, db, None


def _scaled_linear(x, w, b, alpha):
    if x.ndim != 2 or x.device.type != 'cuda':
        y = x.matmul(w.t()) * alpha                      # (host tensors, unusual ranks: the plain composition)
        return y if b is None else y + b
    return _ScaledLinear.apply(x, w, b, alpha)


class Conv2dLayer(torch.nn.Module):
    def __init__(self,
        in_channels,
        out_channels,
        kernel_size,
        bias            = True,
        activation      = 'linear',
        up              = 1,
        down            = 1,
        resample_filter = [1,3,3,1],
        conv_clamp      = None,
        channels_last   = False,
        trainable       = True,
    ):
        super().__init__()
        self.activation = activation
        self.up = up
        self.down = down
        self.conv_clamp = conv_clamp
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        weight = torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=fmt)
        bias = torch.zeros([out_channels]) if bias else None
        if trainable:
            self.weight = torch.nn.Parameter(weight)
            self.bias = torch.nn.Parameter(bias) if bias is not None else None
        else:
            self.register_buffer('weight', weight)
            if bias is not None:
                self.register_buffer('bias', bias)
            else:
                self.bias = None

    def forward(self, x, gain=1, prefiltered=False, then_lowpass_of=None):
        """`then_lowpass_of` (a down-sampling Conv2dLayer that consumes the result next, with `prefiltered=True`): also apply THAT layer's low-pass, as one
        first-order Function with this layer's convolution (ops/conv_bias_act.conv2d_bias_act_fir); the caller has checked fir_fusable"""
        b = self.bias               # fp32 as stored: the fused epilogue takes fp32, the unfused tail casts
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        if then_lowpass_of is not None:
            nxt = then_lowpass_of
            assert self.up == 1 and self.down == 1 and nxt.up == 1 and nxt.down > 1
            return conv_bias_act.conv2d_bias_act_fir(x, self.weight, b, nxt.resample_filter, conv2d_resample.lowpass_padding(nxt.resample_filter, nxt.down, nxt.padding),
                                                     padding=self.padding, act=self.activation, gain=self.act_gain * gain, clamp=clamp, wgain=self.weight_gain)
        # conv2d_resample + bias_act (reference :179-184); the bias_act rides in the convolution kernel's epilogue when it can
        tail = dict(b=b, act=self.activation, alpha=None, gain=self.act_gain * gain, clamp=clamp)
        if conv2d_gradfix.is_mixed(x, self.weight) and x.device.type == 'cuda':
            # 16-bit block: hand the fp32 parameter over; `w * weight_gain`, the cast and the operand layout are one (cached) kernel
            return conv2d_resample.conv2d_resample(x=x, w=self.weight, f=self.resample_filter, up=self.up, down=self.down, padding=self.padding,
                                                   flip_weight=(self.up == 1), bias_act_tail=tail, wgain=self.weight_gain, prefiltered=prefiltered)
        w = self.weight * self.weight_gain
        return conv2d_resample.conv2d_resample(x=x, w=w.to(x.dtype), f=self.resample_filter, up=self.up, down=self.down,
                                               padding=self.padding, flip_weight=(self.up == 1), bias_act_tail=tail, prefiltered=prefiltered)


class MappingNetwork(torch.nn.Module):
    def __init__(self,
        z_dim           = None,     # 0 = no latent
        c_dim           = None,     # 0 = no label
        w_dim           = None,
        num_ws          = None,     # None = do not broadcast
        num_layers      = 8,
        embed_features  = None,     # None = w_dim
        layer_features  = None,     # None = w_dim
        activation      = 'lrelu',
        lr_multiplier   = 0.01,
        w_avg_beta      = 0.995,    # None = do not track
    ):
        assert z_dim is not None and c_dim is not None and w_dim is not None
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim = z_dim, c_dim, w_dim
        self.num_ws, self.num_layers, self.w_avg_beta = num_ws, num_layers, w_avg_beta
        if embed_features is None:
            embed_features = w_dim
        if c_dim == 0:
            embed_features = 0
        if layer_features is None:
            layer_features = w_dim
        widths = [z_dim + embed_features] + [layer_features] * (num_layers - 1) + [w_dim]
        if c_dim > 0:
            self.embed = FullyConnectedLayer(c_dim, embed_features)
        for idx in range(num_layers):
            setattr(self, f'fc{idx}', FullyConnectedLayer(widths[idx], widths[idx + 1], activation=activation, lr_multiplier=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer('w_avg', torch.zeros([w_dim]))

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, skip_w_avg_update=False):
        """z [N, z_dim], c [N, c_dim] -> w [N, num_ws, w_dim] (reference :232-268): normalised latent (and embedded label) through the FC stack,
        the running average of w tracked while training, the result repeated per synthesis layer, optionally pulled towards the average"""
        parts = []
        if self.z_dim > 0:
            misc.assert_shape(z, [None, self.z_dim])
            parts.append(normalize_2nd_moment(z.to(torch.float32)))
        if self.c_dim > 0:
            misc.assert_shape(c, [None, self.c_dim])
            parts.append(normalize_2nd_moment(self.embed(c.to(torch.float32))))
        w = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
        for idx in range(self.num_layers):
            w = getattr(self, f'fc{idx}')(w)
        track = self.w_avg_beta is not None and self.training and not skip_w_avg_update
        if track:
            # `w_avg_rounds` > 1 (set by StepEngine while it evaluates several accumulation rounds in one pass): the batch is [round 0; round 1; ...]
            # and the average advances once per round, in order, exactly as it does over separate calls
            for part in w.detach().chunk(max(int(getattr(self, 'w_avg_rounds', 1)), 1)):
                self.w_avg.copy_(torch.lerp(part.mean(dim=0), self.w_avg, self.w_avg_beta))
        if self.num_ws is not None:
            w = w.unsqueeze(1).repeat([1, self.num_ws, 1])
        if truncation_psi == 1:
            return w
        assert self.w_avg_beta is not None
        if self.num_ws is None or truncation_cutoff is None:
            return torch.lerp(self.w_avg, w, truncation_psi)
        w[:, :truncation_cutoff] = torch.lerp(self.w_avg, w[:, :truncation_cutoff], truncation_psi)
        return w


class SynthesisLayer(torch.nn.Module):
    def __init__(self,
        in_channels     = None,
        out_channels    = None,
        w_dim           = None,
        resolution      = None,
        kernel_size     = 3,
        up              = 1,
        use_noise       = True,
        activation      = 'lrelu',
        resample_filter = (1,3,3,1),
        conv_clamp      = None,
        channels_last   = False,
    ):
        assert None not in (in_channels, out_channels, w_dim, resolution)
        super().__init__()
        self.resolution = resolution
        self.up = up
        self.use_noise = use_noise
        self.activation = activation
        self.conv_clamp = conv_clamp
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(list(resample_filter)))
        self.padding = kernel_size // 2
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=fmt))
        if use_noise:
            self.register_buffer('noise_const', torch.randn([resolution, resolution]))
            self.noise_strength = torch.nn.Parameter(torch.zeros([]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))

    def forward(self, x, w, noise_mode='random', fused_modconv=True, gain=1, x_sole_consumer=False, post_scale=None, styles=None, x_premodulated=False):
        """Extensions for the block that owns the layer:
        `x_sole_consumer`: the caller guarantees that this layer is the only reader of x (a block's conv1 after its conv0), which lets the fused
        training path chain the two layers' backward heads (ops/modconv.modconv_bias_act);
        `post_scale` [N, Cout] (inference passes): return the output times post_scale[n, c] -- the NEXT layer's style modulation, applied in this layer's
        last kernel when that is the fused low-pass tail -- to a reader that passes `x_premodulated=True` together with its `styles`."""
        assert noise_mode in ['random', 'const', 'none']
        misc.assert_shape(x, [None, self.weight.shape[1], self.resolution // self.up, self.resolution // self.up])
        assert not (post_scale is not None or x_premodulated) or not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or self.weight.requires_grad))
        if styles is None:
            styles = self.affine(w)
        noise = None
        if self.use_noise and noise_mode == 'random':
            noise = torch.randn([x.shape[0], 1, self.resolution, self.resolution], device=x.device) * self.noise_strength
        if self.use_noise and noise_mode == 'const':
            noise = self.noise_const * self.noise_strength
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        no_grad = not (torch.is_grad_enabled() and (x.requires_grad or styles.requires_grad or self.weight.requires_grad or self.bias.requires_grad))
        if (no_grad and self.up == 1 and self.activation in ('linear', 'relu', 'lrelu') and x.dtype == torch.bfloat16
                and conv2d_gradfix.epilogue_fusable(x)):
            # inference-only pass (e.g. the generator inside the discriminator phases): demodulation, noise, bias, activation,
            # gain and clamp all ride in the convolution kernel's epilogue -- one read of x*s, one write of y
            with torch.no_grad():
                dcoefs = demod_coefficients(self.weight, styles, x.dtype)
                xs = x if x_premodulated else modulate.scale_nc(x, styles)
                spec = bias_act.activation_funcs[self.activation]
                epi = conv2d_gradfix.Epilogue(oscale=dcoefs, noise=noise, bias=self.bias, act=self.activation, alpha=spec.def_alpha,
                                              gain=self.act_gain * gain, clamp=(clamp if clamp is not None else -1))
                y = conv2d_gradfix._conv_forward(xs, self.weight, (1, 1), (self.padding, self.padding), epi=epi)
                return y if post_scale is None else modulate.scale_nc(y, post_scale)
        assert not x_premodulated, 'x_premodulated: the caller checks premodulated_input_ok() first'
        if modconv.usable(x, self.weight, self.activation, self.up):
            # training pass, first order: same fused epilogue, plus a one-pass backward head (torch_utils/ops/modconv.py)
            return modconv.modconv_bias_act(x, self.weight, styles, demod_coefficients(self.weight, styles, x.dtype), noise, self.bias,
                                            padding=self.padding, act=self.activation, gain=self.act_gain * gain, clamp=clamp,
                                            x_sole_consumer=x_sole_consumer)
        if (self.up == 2 and modconv.enabled and x.device.type == 'cuda' and x.dtype == torch.bfloat16
                and self.activation in ('linear', 'relu', 'lrelu')):
            # up-sampling layer, first-order training pass: x * s -> transposed convolution (one multi-phase launch) -> low-pass whose kernel
            # epilogue carries demodulation, noise, bias, activation, gain and clamp (falls back to the composition inside conv2d_resample)
            spec = bias_act.activation_funcs[self.activation]
            tail = dict(dcoefs=demod_coefficients(self.weight, styles, x.dtype), noise=noise, b=self.bias, act=self.activation, alpha=spec.def_alpha,
                        gain=self.act_gain * gain, clamp=(clamp if clamp is not None else -1), post=post_scale)
            return conv2d_resample.conv2d_resample(x=modulate.scale_nc(x, styles), w=self.weight, f=self.resample_filter, up=2,
                                                   padding=self.padding, flip_weight=False, fir_tail=tail)
        x = modulated_conv2d(x=x, weight=self.weight, styles=styles, noise=noise, up=self.up, padding=self.padding,
                             resample_filter=self.resample_filter, flip_weight=(self.up == 1), fused_modconv=fused_modconv)
        x = bias_act.bias_act(x, self.bias.to(x.dtype), act=self.activation, gain=self.act_gain * gain, clamp=clamp)
        return x if post_scale is None else modulate.scale_nc(x, post_scale)

    def premodulated_input_ok(self, dtype, device, records_graph):
        """will forward(x_premodulated=True) take the inference branch that skips `x * styles`?  (what the caller checks before it asks the previous layer
        for post_scale; `records_graph`: does anything that enters the pass require grad while grad mode is on)"""
        return (not records_graph and not self.bias.requires_grad and self.up == 1 and self.activation in ('linear', 'relu', 'lrelu')
                and dtype == torch.bfloat16 and device.type == 'cuda')


class ToRGBLayer(torch.nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None, channels_last=False):
        super().__init__()
        self.conv_clamp = conv_clamp
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=fmt))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))

    def forward(self, x, w, fused_modconv=True, styles=None):
        """`styles` (extension): this layer's `affine(w) * weight_gain`, when the network has evaluated the affine maps of the whole pass at once"""
        if styles is None:
            styles = self.affine(w) * self.weight_gain
        if torgb.usable(x, self.weight):        # first-order passes: x streams once through per-sample weights (ops/torgb.py); fp32 planar result
            wmod = self.weight.reshape(1, self.weight.shape[0], -1) * styles.to(torch.float32).unsqueeze(1)
            return torgb.torgb(x, wmod, self.bias, clamp=self.conv_clamp)
        if conv_bias_act.fusable(x, self.weight, 'linear'):      # modulation pass, then 1x1 convolution with bias + clamp in its epilogue
            xs = modulate.scale_nc(x, styles)
            wt = self.weight if conv2d_gradfix.is_mixed(x, self.weight) else self.weight.to(x.dtype)
            return conv_bias_act.conv2d_bias_act(xs, wt, self.bias, act='linear', clamp=self.conv_clamp)
        x = modulated_conv2d(x=x, weight=self.weight, styles=styles, demodulate=False, fused_modconv=fused_modconv)
        return bias_act.bias_act(x, self.bias.to(x.dtype), clamp=self.conv_clamp)


Synthlayerkwargs = generators.make_dataclass_from_init(SynthesisLayer.__init__, 'Synthlayerkwargs', None)


def _attention_module(channels):
    from ..biggan.layers import Attention
    return Attention(channels)


class SynthesisBlock(torch.nn.Module):
    def __init__(self,
        in_channels         = None,         # 0 = first block
        out_channels        = None,
        w_dim               = None,
        resolution          = None,
        img_channels        = None,
        is_last             = None,
        architecture        = 'skip',       # 'orig', 'skip', 'resnet'
        resample_filter     = (1,3,3,1),
        conv_clamp          = None,
        use_fp16            = False,        # run this block in LOW_PRECISION
        fp16_channels_last  = False,        # kept for config parity; reduced-precision blocks are always channel-minor here
        attention           = False,        # self-attention at the end of the block
        layer_kwargs        = Synthlayerkwargs(),
    ):
        assert None not in (in_channels, out_channels, w_dim, resolution, img_channels, is_last)
        assert architecture in ['orig', 'skip', 'resnet']
        super().__init__()
        self.in_channels = in_channels
        self.w_dim = w_dim
        self.resolution = resolution
        self.img_channels = img_channels
        self.is_last = is_last
        self.architecture = architecture
        self.use_fp16 = use_fp16
        self.channels_last = bool(use_fp16)
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(list(resample_filter)))
        self.num_conv = 0
        self.num_torgb = 0
        self.attention = _attention_module(out_channels) if attention else None

        if in_channels == 0:
            self.const = torch.nn.Parameter(torch.randn([out_channels, resolution, resolution]))
        lk = dict(layer_kwargs.items()) if layer_kwargs is not None else {}
        lk.update(in_channels=in_channels, out_channels=out_channels, w_dim=w_dim, resolution=resolution, up=2,
                  resample_filter=resample_filter, conv_clamp=conv_clamp, channels_last=self.channels_last)
        if in_channels != 0:
            self.conv0 = SynthesisLayer(**lk)
            self.num_conv += 1
        lk.update(in_channels=out_channels, up=1, resample_filter=[1, 3, 3, 1])
        self.conv1 = SynthesisLayer(**lk)
        self.num_conv += 1
        if is_last or architecture == 'skip':
            self.torgb = ToRGBLayer(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp, channels_last=self.channels_last)
            self.num_torgb += 1
        if in_channels != 0 and architecture == 'resnet':
            self.skip = Conv2dLayer(in_channels, out_channels, kernel_size=1, bias=False, up=2,
                                    resample_filter=resample_filter, channels_last=self.channels_last)

    def forward(self, x, img, ws, force_fp32=False, fused_modconv=None, styles=None, **layer_kwargs):
        """`styles` (extension): (conv0's, conv1's, torgb's) styles or None each, from SynthesisNetwork's one-launch style bank"""
        s0, s1, srgb = styles if styles is not None else (None, None, None)
        misc.assert_shape(ws, [None, self.num_conv + self.num_torgb, self.w_dim])
        w_iter = iter(ws.unbind(dim=1))
        reduced = self.use_fp16 and not force_fp32
        dtype = LOW_PRECISION if reduced else torch.float32
        fmt = torch.channels_last if reduced else torch.contiguous_format
        if fused_modconv is None:
            fused_modconv = (not self.training) and (dtype == torch.float32 or int(x.shape[0]) == 1)

        lk = dict(fused_modconv=fused_modconv, **layer_kwargs)
        first = self.in_channels == 0
        if first:       # the learned constant, one copy per sample
            x = self.const.to(dtype=dtype, memory_format=fmt).unsqueeze(0).repeat([ws.shape[0], 1, 1, 1])
        else:
            misc.assert_shape(x, [None, self.in_channels, self.resolution // 2, self.resolution // 2])
            x = x.to(dtype=dtype, memory_format=fmt)

        # main path: [conv0 (up-sampling)] -> conv1; 'resnet' adds a 1x1 up-sampling shortcut, both branches scaled by sqrt(1/2)
        residual = (not first) and self.architecture == 'resnet'
        shortcut = self.skip(x, gain=np.sqrt(0.5)) if residual else None
        sole = not first        # conv0's output is read by conv1 and by nothing else
        g1 = np.sqrt(0.5) if residual else 1
        records_graph = torch.is_grad_enabled() and (x.requires_grad or ws.requires_grad or self.conv1.weight.requires_grad or (sole and self.conv0.weight.requires_grad))
        if sole and premodulate and self.conv1.premodulated_input_ok(dtype, x.device, records_graph):
            # inference pass: conv0's last kernel hands conv1 its input already multiplied by conv1's styles (no `x * styles` pass in between)
            w0, w1 = next(w_iter), next(w_iter)
            styles1 = self.conv1.affine(w1) if s1 is None else s1
            x = self.conv0(x, w0, post_scale=styles1, styles=s0, **lk)
            x = self.conv1(x, w1, gain=g1, styles=styles1, x_premodulated=True, **lk)
        else:
            if not first:
                x = self.conv0(x, next(w_iter), styles=s0, **lk)
            x = self.conv1(x, next(w_iter), gain=g1, x_sole_consumer=sole, styles=s1, **lk)
        if residual:
            x = shortcut + x     # out of place: both summands are outputs of fused conv + activation ops, which keep them for their backward
        if self.attention is not None:
            x = self.attention(x.to(torch.float32)).to(dtype)

        # image branch: the running fp32 image is up-sampled, and 'skip' (or the last block of any architecture) adds this block's RGB
        if img is not None:
            misc.assert_shape(img, [None, self.img_channels, self.resolution // 2, self.resolution // 2])
            img = upfirdn2d.upsample2d(img, self.resample_filter)
        if self.is_last or self.architecture == 'skip':
            rgb = self.torgb(x, next(w_iter), fused_modconv=fused_modconv, styles=srgb).to(dtype=torch.float32, memory_format=torch.contiguous_format)
            img = rgb if img is None else img.add_(rgb)
        assert x.dtype == dtype and (img is None or img.dtype == torch.float32)
        return x, img


Synthblockkwargs = generators.make_dataclass_from_init(SynthesisBlock.__init__, 'Synthblockkwargs', None)


class SynthesisNetwork(torch.nn.Module):
    def __init__(self,
        w_dim           = None,
        img_resolution  = None,
        img_channels    = None,
        channel_base    = 32768,
        channel_max     = 512,
        num_fp16_res    = 0,        # run the N highest resolutions in LOW_PRECISION
        attentions      = (),
        block_kwargs    = Synthblockkwargs(),
    ):
        assert None not in (w_dim, img_resolution, img_channels)
        assert img_resolution >= 4 and img_resolution & (img_resolution - 1) == 0
        super().__init__()
        self.w_dim = w_dim
        self.img_resolution = img_resolution
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.img_channels = img_channels
        self.block_resolutions = [2 ** i for i in range(2, self.img_resolution_log2 + 1)]
        channels = {res: min(channel_base // res, channel_max) for res in self.block_resolutions}
        fp16_resolution = max(2 ** (self.img_resolution_log2 + 1 - num_fp16_res), 8)
        bk = dict(block_kwargs.items()) if block_kwargs is not None else {}
        self.num_ws = 0
        for res in self.block_resolutions:
            bk.update(in_channels=(channels[res // 2] if res > 4 else 0), out_channels=channels[res], w_dim=w_dim, resolution=res,
                      img_channels=img_channels, is_last=(res == img_resolution), use_fp16=(res >= fp16_resolution),
                      attention=(res in attentions))
            block = SynthesisBlock(**bk)
            self.num_ws += block.num_conv
            if res == img_resolution:
                self.num_ws += block.num_torgb
            setattr(self, f'b{res}', block)

    def forward(self, ws, **block_kwargs):
        misc.assert_shape(ws, [None, self.num_ws, self.w_dim])
        ws = ws.to(torch.float32)
        per_block, w_idx = [], 0
        for res in self.block_resolutions:
            block = getattr(self, f'b{res}')
            per_block.append(ws.narrow(1, w_idx, block.num_conv + block.num_torgb))
            w_idx += block.num_conv
        x = img = None
        plan = self.pass_plan(ws.shape[0]) if ws.shape[0] > 1 else None
        tail, chunk = plan if plan is not None else (0, ws.shape[0])
        head = len(self.block_resolutions) - tail
        bank = self._style_bank(ws)       # {res: (conv0's, conv1's, torgb's styles)} from one launch, or None: every layer evaluates its own affine map
        for res, cur_ws in zip(self.block_resolutions[:head], per_block[:head]):
            x, img = getattr(self, f'b{res}')(x, img, cur_ws, styles=(bank[res] if bank else None), **block_kwargs)
        if tail > 0:        # the highest-resolution blocks over slices of the batch (pass_plan), their images concatenated
            imgs = []
            for i in range(0, ws.shape[0], chunk):
                xc, ic = x.narrow(0, i, chunk), (img.narrow(0, i, chunk) if img is not None else None)
                for res, cur_ws in zip(self.block_resolutions[head:], per_block[head:]):
                    st = tuple(None if s is None else s.narrow(0, i, chunk) for s in bank[res]) if bank else None
                    xc, ic = getattr(self, f'b{res}')(xc, ic, cur_ws.narrow(0, i, chunk), styles=st, **block_kwargs)
                imgs.append(ic)
            img = misc.cat0(imgs)
        return img

    def _style_bank(self, ws):
        """every layer's styles of this pass from ONE launch (torch_utils/ops/grouped_gemm.py) instead of one addmm per layer (reference :333, :397), its
        backward two launches instead of two GEMMs and a reduction per layer.  First-order passes and inference only (the switch the fused synthesis
        layers obey: trainers turn it off for phases that differentiate twice); SBG_STYLE_BANK=0 restores the per-layer calls."""
        if not (style_bank_enabled and ws.is_cuda and ws.dtype == torch.float32 and (modconv.enabled or not torch.is_grad_enabled())):
            return None
        layers, where, w_idx = [], [], 0
        for res in self.block_resolutions:
            block = getattr(self, f'b{res}')
            has0 = block.in_channels != 0
            entries = [(block.conv0.affine, w_idx, 1.0) if has0 else None, (block.conv1.affine, w_idx + (1 if has0 else 0), 1.0),
                       (block.torgb.affine, w_idx + block.num_conv, float(block.torgb.weight_gain)) if block.num_torgb else None]
            for pos, e in enumerate(entries):
                if e is None:
                    continue
                fc, slot, post = e
                if fc.activation != 'linear' or fc.bias is None or fc.weight.dtype != torch.float32 or not fc.weight.is_contiguous():
                    return None
                layers.append((slot, fc.weight, fc.bias, float(fc.weight_gain) * post, float(fc.bias_gain) * post))
                where.append((res, pos))
            w_idx += block.num_conv
        styles = grouped_gemm.style_bank(ws, layers)
        bank = {res: [None, None, None] for res in self.block_resolutions}
        for (res, pos), s in zip(where, styles):
            bank[res][pos] = s
        return {res: tuple(v) for res, v in bank.items()}

    pass_bytes_limit = 1 << 31      # the op layer addresses tensors below 2 GiB

    def pass_plan(self, n):
        """(k, chunk): the k highest-resolution blocks run over slices of `chunk` samples so that no tensor of a pass over n samples reaches
        `pass_bytes_limit`; (0, n) = no slicing; None = not possible.  Blocks treat samples independently (no attention in the sliced blocks); the
        counterpart of Discriminator.pass_plan for the generator's trailing blocks."""
        cache = self.__dict__.setdefault('_pass_plans', {})
        key = (n, self.pass_bytes_limit)
        if key not in cache:
            blocks = [getattr(self, f'b{res}') for res in self.block_resolutions]
            peaks = [int(b.conv1.weight.shape[0]) * (b.resolution + 1) ** 2 * (2 if b.use_fp16 else 12) for b in blocks]
            k = 0
            while k < len(peaks) - 1 and n * peaks[-1 - k] >= self.pass_bytes_limit:
                k += 1
            plan = (0, n)
            if k > 0:
                plan = None
                if not any(b.attention is not None for b in blocks[len(blocks) - k:]):
                    for chunk in range(n // 2, 0, -1):
                        if n % chunk == 0 and chunk * max(peaks[len(peaks) - k:]) < self.pass_bytes_limit:
                            plan = (k, chunk)
                            break
            cache[key] = plan
        return cache[key]


Mappingkwargs = generators.make_dataclass_from_init(MappingNetwork.__init__, 'Mappingkwargs', None)
Synthesiskwargs = generators.make_dataclass_from_init(SynthesisNetwork.__init__, 'Synthesiskwargs', None)


@generators.add_to_registry("sg2_classic")
class Generator(torch.nn.Module):
    def __init__(self,
        z_dim               = 128,
        c_dim               = None,
        w_dim               = 128,
        img_resolution      = None,
        img_channels        = None,
        attentions          = (),
        mapping_kwargs      = Mappingkwargs(),
        synthesis_kwargs    = Synthesiskwargs(),
    ):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim = z_dim, c_dim, w_dim
        self.img_resolution, self.img_channels = img_resolution, img_channels
        sk = dict(synthesis_kwargs.items()) if synthesis_kwargs is not None else {}
        sk.update(w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels, attentions=attentions)
        self.synthesis = SynthesisNetwork(**sk)
        self.num_ws = self.synthesis.num_ws
        mk = dict(mapping_kwargs.items()) if mapping_kwargs is not None else {}
        mk.update(z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=self.num_ws)
        self.mapping = MappingNetwork(**mk)
        # several accumulation rounds may be evaluated in one pass when nothing but the mapping network's running average carries state across
        # forward calls (the attention blocks' spectral-norm power iterations do) -- StepEngine._rounds_per_pass
        self.rounds_mergeable = len(tuple(attentions)) == 0

    def peak_activation_bytes(self):
        """bytes per sample of the largest tensor a forward pass creates (the [C, res + 1, res + 1] output of an up-sampling convolution
        before its low-pass); see Discriminator.peak_activation_bytes"""
        blocks = [getattr(self.synthesis, f'b{res}') for res in self.synthesis.block_resolutions]
        return max(int(b.conv1.weight.shape[0]) * (b.resolution + 1) ** 2 * (2 if b.use_fp16 else 12) for b in blocks)

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
        return self.synthesis(ws, **synthesis_kwargs)


# ----------------------------------------------------------------------------------------------------------------
# DCGAN (plumbing config `configs/dcgan.yaml`: stock torch.nn layers, runs on CPU eager -- reference :569-606)

def _dcgan_up(cin, cout, k, stride, pad, last=False):
    conv = torch.nn.ConvTranspose2d(cin, cout, k, stride, pad, bias=False)
    return [conv, torch.nn.Tanh()] if last else [conv, torch.nn.BatchNorm2d(cout), torch.nn.ReLU(True)]


class Generator_dcgan(torch.nn.Module):
    """z -> [z_dim, 1, 1] -> M x M -> four stride-2 transposed 4x4 convolutions -> 16M x 16M RGB in [-1, 1]"""

    def __init__(self, z_dim, M):
        super().__init__()
        self.z_dim = z_dim
        layers = _dcgan_up(z_dim, 1024, M, 1, 0)
        for cin, cout in [(1024, 512), (512, 256), (256, 128)]:
            layers += _dcgan_up(cin, cout, 4, 2, 1)
        layers += _dcgan_up(128, 3, 4, 2, 1, last=True)
        self.main = torch.nn.Sequential(*layers)

    def forward(self, z, c, noise_mode=None):
        return self.main(z.view(-1, self.z_dim, 1, 1))


@generators.add_to_registry("cnn32_dcgan")
class Generator32_dcgan(Generator_dcgan):
    def __init__(self, z_dim, c_dim, img_resolution, *args, **kwargs):
        super().__init__(z_dim, M=2)
        self.c_dim = c_dim
        self.img_resolution = img_resolution


@generators.add_to_registry("cnn48_dcgan")
class Generator48_dcgan(Generator_dcgan):
    def __init__(self, z_dim):
        super().__init__(z_dim, M=4)


# ----------------------------------------------------------------------------------------------------------------
# BigGAN (reference :720-937): spectral-norm convolutions, class-conditional batch norm, optional self-attention

def G_arch(ch=64, attention='64', ksize='333333', dilation='111111'):
    """channel / resolution schedule per output resolution (reference :720-753)"""
    att = [int(item) for item in attention.split('_')]
    table = {512: ([16, 16, 8, 8, 4, 2, 1], [16, 8, 8, 4, 2, 1, 1]), 256: ([16, 16, 8, 8, 4, 2], [16, 8, 8, 4, 2, 1]),
             128: ([16, 16, 8, 4, 2], [16, 8, 4, 2, 1]), 64: ([16, 16, 8, 4], [16, 8, 4, 2]), 32: ([4, 4, 4], [4, 4, 4])}
    arch = {}
    for res, (cin, cout) in table.items():
        resolutions = [2 ** i for i in range(3, int(np.log2(res)) + 1)]
        arch[res] = {'in_channels': [ch * m for m in cin], 'out_channels': [ch * m for m in cout], 'upsample': [True] * len(cin),
                     'resolution': resolutions, 'attention': {r: (r in att) for r in resolutions}}
    return arch


@generators.add_to_registry("big_gan")
class BigGAnGenerator(torch.nn.Module):
    def __init__(self, G_ch=64, z_dim=128, c_dim=10, bottom_width=4, img_resolution=128,
                 G_kernel_size=3, G_attn='64', n_classes=10,
                 num_G_SVs=1, num_G_SV_itrs=1,
                 G_shared=True, shared_dim=0, hier=False,
                 cross_replica=False, mybn=False,
                 G_activation='relu',
                 BN_eps=1e-5, SN_eps=1e-12, G_mixed_precision=False, G_fp16=False,
                 G_init='ortho',
                 G_param='SN', norm_style='bn',
                 **kwargs):
        super().__init__()
        import functools
        from ..biggan import layers
        self.c_dim, self.ch, self.z_dim = c_dim, G_ch, z_dim
        self.bottom_width, self.img_resolution = bottom_width, img_resolution
        self.kernel_size, self.attention, self.n_classes = G_kernel_size, G_attn, n_classes
        self.G_shared = G_shared
        self.shared_dim = shared_dim if shared_dim > 0 else z_dim
        self.hier, self.cross_replica, self.mybn = hier, cross_replica, mybn
        assert G_activation == 'relu'
        self.activation = layers.ReLU()
        self.init, self.G_param, self.norm_style = G_init, G_param, norm_style
        self.BN_eps, self.SN_eps, self.fp16 = BN_eps, SN_eps, G_fp16
        self.arch = G_arch(self.ch, self.attention)[img_resolution]
        if self.hier:
            self.num_slots = len(self.arch['in_channels']) + 1
            self.z_chunk_size = self.z_dim // self.num_slots
            self.z_dim = self.z_chunk_size * self.num_slots
        else:
            self.num_slots, self.z_chunk_size = 1, 0

        assert self.G_param == 'SN', "G_param='SN' (the shipped configuration) is implemented"
        self.which_conv = functools.partial(layers.SNConv2d, kernel_size=3, padding=1, num_svs=num_G_SVs, num_itrs=num_G_SV_itrs, eps=self.SN_eps)
        self.which_linear = functools.partial(layers.SNLinear, num_svs=num_G_SVs, num_itrs=num_G_SV_itrs, eps=self.SN_eps)
        self.which_embedding = torch.nn.Embedding
        bn_linear = functools.partial(self.which_linear, bias=False) if self.G_shared else self.which_embedding
        self.which_bn = functools.partial(layers.ccbn, which_linear=bn_linear, cross_replica=self.cross_replica, mybn=self.mybn,
                                          input_size=(self.shared_dim + self.z_chunk_size if self.G_shared else self.n_classes),
                                          norm_style=self.norm_style, eps=self.BN_eps)
        self.shared = self.which_embedding(n_classes, self.shared_dim) if G_shared else layers.identity()
        self.linear = self.which_linear(self.z_dim // self.num_slots, self.arch['in_channels'][0] * (self.bottom_width ** 2))
        blocks = []
        for index in range(len(self.arch['out_channels'])):
            stage = [layers.GBlock(in_channels=self.arch['in_channels'][index], out_channels=self.arch['out_channels'][index],
                                   which_conv=self.which_conv, which_bn=self.which_bn, activation=self.activation,
                                   upsample=(layers.nearest_upsample2x if self.arch['upsample'][index] else None))]
            if self.arch['attention'][self.arch['resolution'][index]]:
                stage.append(layers.Attention(self.arch['out_channels'][index], self.which_conv))
            blocks.append(torch.nn.ModuleList(stage))
        self.blocks = torch.nn.ModuleList(blocks)
        self.output_layer = torch.nn.Sequential(layers.bn(self.arch['out_channels'][-1], cross_replica=self.cross_replica, mybn=self.mybn),
                                                self.activation, self.which_conv(self.arch['out_channels'][-1], 3))
        self.init_weights()

    def init_weights(self):
        self.param_count = 0
        for module in self.modules():
            if isinstance(module, (torch.nn.Conv2d, torch.nn.Linear, torch.nn.Embedding)):
                if self.init == 'ortho':
                    torch.nn.init.orthogonal_(module.weight)
                elif self.init == 'N02':
                    torch.nn.init.normal_(module.weight, 0, 0.02)
                elif self.init in ['glorot', 'xavier']:
                    torch.nn.init.xavier_uniform_(module.weight)
                self.param_count += sum(p.data.nelement() for p in module.parameters())

    def forward(self, z, c, noise_mode='random'):
        y = torch.argmax(c, dim=1)      # the reference feeds class indices (G_shared=False: embedding lookup inside ccbn)
        if self.hier:
            zs = torch.split(z, self.z_chunk_size, 1)
            z = zs[0]
            ys = [torch.cat([y, item], 1) for item in zs[1:]]
        else:
            ys = [y] * len(self.blocks)
        h = self.linear(z)
        h = h.view(h.size(0), -1, self.bottom_width, self.bottom_width)
        for index, blocklist in enumerate(self.blocks):
            for block in blocklist:
                h = block(h, ys[index])
        return torch.tanh(self.output_layer(h))
