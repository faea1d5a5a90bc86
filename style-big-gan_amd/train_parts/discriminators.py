"""Discriminators (host-side modules over the HIP op layer).

Mirror of the registry surface and module tree of the reference's ``train_parts/discriminators.py``: registry
``discriminators`` with ``@discriminators.add_to_registry("sg2_classic" | "cnn32_dcgan" | "cnn48_dcgan" | "big_gan")``; for
StyleGAN2 ``DiscriminatorBlock`` (:211), ``MinibatchStdLayer`` (:307), ``DiscriminatorEpilogue`` (:332) and
``Discriminator`` (:403) keep constructor arguments, parameter / buffer names and forward semantics, so state_dicts
interchange with the reference.  Reduced-precision blocks run in ``generators.LOW_PRECISION`` (bf16 by default),
channel-minor, on the hand-written kernels behind ``torch_utils.ops``.
"""
import numpy as np
import torch

from .. import utils
from ..torch_utils import misc
from ..torch_utils.ops import conv2d_gradfix, conv_bias_act, fromrgb, upfirdn2d
from . import generators as _g
from .generators import Conv2dLayer, FullyConnectedLayer, MappingNetwork

discriminators = utils.ClassRegistry()


def _attention_module(channels):
    from ..biggan.layers import Attention
    return Attention(channels)


class DiscriminatorBlock(torch.nn.Module):
    def __init__(self,
        in_channels         = None,         # 0 = first block
        tmp_channels        = None,
        out_channels        = None,
        resolution          = None,
        img_channels        = None,
        first_layer_idx     = None,
        architecture        = 'resnet',     # 'orig', 'skip', 'resnet'
        attention           = False,
        activation          = 'lrelu',
        resample_filter     = (1,3,3,1),
        conv_clamp          = None,
        use_fp16            = False,        # run this block in LOW_PRECISION
        fp16_channels_last  = False,        # kept for config parity (reduced-precision blocks are always channel-minor)
        freeze_layers       = 0,            # Freeze-D
    ):
        assert None not in (in_channels, tmp_channels, out_channels, resolution, img_channels, first_layer_idx)
        assert in_channels in [0, tmp_channels]
        assert architecture in ['orig', 'skip', 'resnet']
        super().__init__()
        self.in_channels = in_channels
        self.resolution = resolution
        self.img_channels = img_channels
        self.first_layer_idx = first_layer_idx
        self.architecture = architecture
        self.use_fp16 = use_fp16
        self.channels_last = bool(use_fp16)
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(list(resample_filter)))
        self.attention = _attention_module(out_channels) if attention else None

        self.num_layers = 0

        def next_trainable():
            trainable = (self.first_layer_idx + self.num_layers) >= freeze_layers
            self.num_layers += 1
            return trainable

        common = dict(activation=activation, conv_clamp=conv_clamp, channels_last=self.channels_last)
        if in_channels == 0 or architecture == 'skip':
            self.fromrgb = Conv2dLayer(img_channels, tmp_channels, kernel_size=1, trainable=next_trainable(), **common)
        self.conv0 = Conv2dLayer(tmp_channels, tmp_channels, kernel_size=3, trainable=next_trainable(), **common)
        self.conv1 = Conv2dLayer(tmp_channels, out_channels, kernel_size=3, down=2, trainable=next_trainable(),
                                 resample_filter=resample_filter, **common)
        if architecture == 'resnet':
            self.skip = Conv2dLayer(tmp_channels, out_channels, kernel_size=1, bias=False, down=2, trainable=next_trainable(),
                                    resample_filter=resample_filter, channels_last=self.channels_last)

    def forward(self, x, img, force_fp32=False):
        reduced = self.use_fp16 and not force_fp32
        dtype = _g.LOW_PRECISION if reduced else torch.float32
        fmt = torch.channels_last if reduced else torch.contiguous_format

        if x is not None:
            misc.assert_shape(x, [None, self.in_channels, self.resolution, self.resolution])
            x = x.to(dtype=dtype, memory_format=fmt)

        if self.in_channels == 0 or self.architecture == 'skip':
            misc.assert_shape(img, [None, self.img_channels, self.resolution, self.resolution])
            fr = self.fromrgb
            if (self.architecture != 'skip' and fr.up == 1 and fr.down == 1 and fromrgb.usable(img, fr.weight, fr.activation, dtype)):
                # first-order passes: the fp32 image streams once through the 1x1 layer (ops/fromrgb.py), no cast / layout pass, no padded GEMM
                y = fromrgb.fromrgb(img, fr.weight, fr.bias, fr.weight_gain, fr.activation, gain=fr.act_gain, clamp=fr.conv_clamp, out_dtype=dtype)
            else:
                img = img.to(dtype=dtype, memory_format=fmt)
                y = self.fromrgb(img)
            x = x + y if x is not None else y
            img = upfirdn2d.downsample2d(img, self.resample_filter) if self.architecture == 'skip' else None

        # conv0 -> conv1 (low-pass + stride 2); 'resnet' adds the 1x1 down-sampling shortcut, both branches scaled by sqrt(1/2)
        residual = self.architecture == 'resnet'
        shortcut = self.skip(x, gain=np.sqrt(0.5)) if residual else None
        c0, c1 = self.conv0, self.conv1
        g1 = np.sqrt(0.5) if residual else 1
        if (conv_bias_act.first_order and x.device.type == 'cuda' and conv2d_gradfix.is_mixed(x, c0.weight) and c0.up == 1 and c0.down == 1 and c1.up == 1
                and c1.down == 2 and c1.weight.shape[2] > 1 and conv_bias_act.fir_fusable(x, c0.weight, c0.activation, c1.resample_filter)):
            # first-order passes: conv0 and the low-pass of conv1 as one Function, whose backward never writes the gradient w.r.t. conv0's output
            x = c1(c0(x, then_lowpass_of=c1), gain=g1, prefiltered=True)
        else:
            x = c1(c0(x), gain=g1)
        if residual:
            x = shortcut + x     # out of place: both summands are outputs of fused conv + activation ops, which keep them for their backward
        if self.attention is not None:
            x = self.attention(x.to(torch.float32)).to(dtype)
        assert x.dtype == dtype
        return x, img


class MinibatchStdLayer(torch.nn.Module):
    """Appends `num_channels` feature maps holding the per-group standard deviation (reference :307-328)."""

    def __init__(self, group_size, num_channels=1):
        super().__init__()
        self.group_size = group_size
        self.num_channels = num_channels

    def forward(self, x):
        N, C, H, W = x.shape
        G = min(int(self.group_size), N) if self.group_size is not None else N
        F = self.num_channels
        if x.device.type == 'cuda' and x.dtype == torch.float32 and x.is_contiguous() and N % G == 0 and C % F == 0:
            return _MinibatchStd.apply(x, G, F)            # one kernel per direction (csrc/mbstd.hip)
        return _minibatch_std_composite(x, G, F)


def _minibatch_std_composite(x, G, F):
    """the layer as tensor ops (reference :316-328): differentiable to any order"""
    N, C, H, W = x.shape
    c = C // F
    y = x.reshape(G, -1, F, c, H, W)
    y = y - y.mean(dim=0)
    y = y.square().mean(dim=0)
    y = (y + 1e-8).sqrt()
    y = y.mean(dim=[2, 3, 4])
    y = y.reshape(-1, F, 1, 1)
    y = y.repeat(G, 1, H, W)
    return torch.cat([x, y], dim=1)


class _MinibatchStd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, G, F):
        from .. import _lib
        N, C, H, W = x.shape
        y = torch.empty([N, C + F, H, W], dtype=torch.float32, device=x.device)
        lib = _lib.load()
        ws = torch.empty([max(lib.sbg_mbstd_workspace(N, C, H * W, G, F), 4) // 4], dtype=torch.float32, device=x.device)
        _lib.check(lib.sbg_mbstd_fwd(_lib.ptr(x), _lib.ptr(y), _lib.ptr(ws), N, C, H * W, G, F, _lib.stream_ptr(x.device)), "sbg_mbstd_fwd")
        ctx.save_for_backward(x)
        ctx.cfg = (G, F)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        G, F = ctx.cfg
        if torch.is_grad_enabled():        # R1 differentiates this gradient again: compose it from differentiable ops on the saved input
            with torch.enable_grad():
                xx = x if x.requires_grad else x.detach().requires_grad_(True)
                (dx,) = torch.autograd.grad(_minibatch_std_composite(xx, G, F), xx, dy, create_graph=True)
            return dx, None, None
        from .. import _lib
        N, C, H, W = x.shape
        dyc = dy.contiguous()
        dx = torch.empty_like(x)
        _lib.check(_lib.load().sbg_mbstd_bwd(_lib.ptr(x), _lib.ptr(dyc), _lib.ptr(dx), N, C, H * W, G, F, _lib.stream_ptr(x.device)), "sbg_mbstd_bwd")
        return dx, None, None


class DiscriminatorEpilogue(torch.nn.Module):
    def __init__(self,
        in_channels         = None,
        cmap_dim            = None,     # 0 = no label
        resolution          = None,
        img_channels        = None,
        architecture        = 'resnet',
        mbstd_group_size    = 4,        # None = entire minibatch
        mbstd_num_channels  = 1,        # 0 = disable
        activation          = 'lrelu',
        conv_clamp          = None,
    ):
        assert None not in (in_channels, cmap_dim, resolution, img_channels)
        assert architecture in ['orig', 'skip', 'resnet']
        super().__init__()
        self.in_channels = in_channels
        self.cmap_dim = cmap_dim
        self.resolution = resolution
        self.img_channels = img_channels
        self.architecture = architecture
        if architecture == 'skip':
            self.fromrgb = Conv2dLayer(img_channels, in_channels, kernel_size=1, activation=activation)
        self.mbstd = MinibatchStdLayer(group_size=mbstd_group_size, num_channels=mbstd_num_channels) if mbstd_num_channels > 0 else None
        self.conv = Conv2dLayer(in_channels + mbstd_num_channels, in_channels, kernel_size=3, activation=activation, conv_clamp=conv_clamp)
        self.fc = FullyConnectedLayer(in_channels * (resolution ** 2), in_channels, activation=activation)
        self.out = FullyConnectedLayer(in_channels, 1 if cmap_dim == 0 else cmap_dim)

    def forward(self, x, img, cmap, force_fp32=False):
        misc.assert_shape(x, [None, self.in_channels, self.resolution, self.resolution])
        x = x.to(dtype=torch.float32, memory_format=torch.contiguous_format)       # the epilogue always runs in fp32 (reference :366-367)
        if self.architecture == 'skip':
            misc.assert_shape(img, [None, self.img_channels, self.resolution, self.resolution])
            x = x + self.fromrgb(img.to(dtype=torch.float32, memory_format=torch.contiguous_format))
        # [group statistics channel] -> 3x3 conv -> dense -> logits (one per sample, or a projection onto the mapped label)
        head = [m for m in (self.mbstd, self.conv) if m is not None]
        for m in head:
            x = m(x)
        logits = self.out(self.fc(x.flatten(1)))
        if self.cmap_dim > 0:
            misc.assert_shape(cmap, [None, self.cmap_dim])
            logits = (logits * cmap).sum(dim=1, keepdim=True) * (1 / np.sqrt(self.cmap_dim))
        assert logits.dtype == torch.float32
        return logits


Mappingkwargs = discriminators.make_dataclass_from_init(MappingNetwork.__init__, 'Mappingkwargs', None)
Discblockkwargs = discriminators.make_dataclass_from_init(DiscriminatorBlock.__init__, 'Discblockkwargs', None)
Discepilogkwargs = discriminators.make_dataclass_from_init(DiscriminatorEpilogue.__init__, 'Discepilogkwargs', None)


@discriminators.add_to_registry("sg2_classic")
class Discriminator(torch.nn.Module):
    def __init__(self,
        c_dim               = None,
        img_resolution      = None,
        img_channels        = None,
        attentions          = (),
        architecture        = 'resnet',
        channel_base        = 32768,
        channel_max         = 512,
        num_fp16_res        = 0,        # run the N highest resolutions in LOW_PRECISION
        conv_clamp          = None,
        cmap_dim            = None,     # None = default
        block_kwargs        = Discblockkwargs(),
        mapping_kwargs      = Mappingkwargs(),
        epilogue_kwargs     = Discepilogkwargs(),
    ):
        assert img_resolution is not None and img_channels is not None
        super().__init__()
        self.c_dim = c_dim
        self.img_resolution = img_resolution
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.img_channels = img_channels
        self.block_resolutions = [2 ** i for i in range(self.img_resolution_log2, 2, -1)]
        channels = {res: min(channel_base // res, channel_max) for res in self.block_resolutions + [4]}
        fp16_resolution = max(2 ** (self.img_resolution_log2 + 1 - num_fp16_res), 8)
        if cmap_dim is None:
            cmap_dim = channels[4]
        if c_dim == 0:
            cmap_dim = 0

        common = dict(img_channels=img_channels, architecture=architecture, conv_clamp=conv_clamp)
        bk = dict(block_kwargs.items()) if block_kwargs is not None else {}
        layer_idx = 0
        for res in self.block_resolutions:
            bk.update(in_channels=(channels[res] if res < img_resolution else 0), tmp_channels=channels[res],
                      out_channels=channels[res // 2], resolution=res, first_layer_idx=layer_idx,
                      use_fp16=(res >= fp16_resolution), attention=(res in attentions))
            bk.update(common)
            block = DiscriminatorBlock(**bk)
            setattr(self, f'b{res}', block)
            layer_idx += block.num_layers
        if c_dim is not None and c_dim > 0:
            mk = dict(mapping_kwargs.items()) if mapping_kwargs is not None else {}
            mk.update(z_dim=0, c_dim=c_dim, w_dim=cmap_dim, num_ws=None, w_avg_beta=None)
            self.mapping = MappingNetwork(**mk)
        ek = dict(epilogue_kwargs.items()) if epilogue_kwargs is not None else {}
        ek.update(in_channels=channels[4], cmap_dim=cmap_dim, resolution=4)
        ek.update(common)
        self.b4 = DiscriminatorEpilogue(**ek)
        # one forward over [fake; real] computes what two forwards compute when nothing in the network carries state across calls (spectral-norm
        # power iterations do: the attention blocks) and the minibatch-std groups are kept apart -- see merged_batch_order()
        self.batch_mergeable = len(tuple(attentions)) == 0

    def merged_batch_order(self, n, segments=2):
        """index list that arranges cat([a_0, ..., a_{S-1}]) (S = `segments` batches of n samples each: the generated and the real half of a
        round, and / or several accumulation rounds) so that the minibatch-std groups of the merged batch are exactly the groups the layer forms on
        each a_s alone (reference MinibatchStdLayer :316-328 groups samples j, j + n/G, j + 2 n/G, ...), or None when the segments cannot be kept
        apart.  With m = n/G groups per segment, position p = r + (S m) k holds sample (r mod m) + m k of segment r div m."""
        S = int(segments)
        mb = self.b4.mbstd
        if mb is None:
            return list(range(S * n))
        G = min(int(mb.group_size), n) if mb.group_size is not None else None
        if G is None or n % G != 0 or min(int(mb.group_size), S * n) != G:
            return None
        m = n // G
        order = []
        for p in range(S * n):
            k, r = divmod(p, S * m)
            order.append((r // m) * n + (r % m) + m * k)
        return order

    def peak_activation_bytes(self):
        """bytes per sample of the largest tensor a forward pass creates (the low-passed input of a block's strided convolution,
        [tmp_channels, res + 1, res + 1]): what bounds the batch of one pass -- the op layer addresses tensors below 2^31 elements / 2 GiB
        (the reference plugins' own limit, upfirdn2d.cpp:22-23).  fp32 blocks count 12 bytes per element: their convolutions run on operands split into
        bf16 parts and concatenated along the reduction axis (six parts of two bytes), and THAT tensor must stay below the limit for the fast kernels"""
        blocks = [getattr(self, f'b{res}') for res in self.block_resolutions]
        return max(int(b.conv1.weight.shape[1]) * (b.resolution + 1) ** 2 * (2 if b.use_fp16 else 12) for b in blocks)

    pass_bytes_limit = 1 << 31      # the op layer addresses tensors below 2 GiB (and, at two bytes per element, below 2^31 elements)

    def pass_plan(self, n):
        """How a pass over n samples stays below `pass_bytes_limit` per tensor: (k, chunk) = the k highest-resolution blocks run over slices of
        `chunk` samples, their outputs are concatenated and the remaining blocks see the whole batch; (0, n) = no slicing; None = not possible.
        The blocks treat samples independently (only the epilogue's minibatch-std layer looks across the batch), so slicing them changes nothing
        but the launch sizes -- while the low-resolution blocks, whose cost is per launch rather than per sample, still run once.  Only for
        architectures whose blocks hand nothing but `x` to the next block ('orig', 'resnet': the image is consumed by the first block)."""
        cache = self.__dict__.setdefault('_pass_plans', {})
        key = (n, self.pass_bytes_limit)
        if key not in cache:
            cache[key] = self._pass_plan(n)
        return cache[key]

    def _pass_plan(self, n):
        blocks = [getattr(self, f'b{res}') for res in self.block_resolutions]
        peaks = [int(b.conv1.weight.shape[1]) * (b.resolution + 1) ** 2 * (2 if b.use_fp16 else 12) for b in blocks]
        k = 0
        while k < len(peaks) and n * peaks[k] >= self.pass_bytes_limit:
            k += 1
        if k == 0:
            return 0, n
        if any(b.architecture == 'skip' or b.attention is not None for b in blocks[:k]):
            return None
        for chunk in range(n // 2, 0, -1):
            if n % chunk == 0 and chunk * max(peaks[:k]) < self.pass_bytes_limit:
                return k, chunk
        return None

    def forward(self, img, c, **block_kwargs):
        x = None
        plan = self.pass_plan(img.shape[0]) if img.shape[0] > 1 else None
        lead, chunk = plan if plan is not None else (0, img.shape[0])
        if lead > 0:
            parts = []
            for part in img.split(chunk):
                xp = None
                for res in self.block_resolutions[:lead]:
                    xp, part = getattr(self, f'b{res}')(xp, part, **block_kwargs)
                parts.append(xp)
            x, img = misc.cat0(parts), None
        for res in self.block_resolutions[lead:]:
            x, img = getattr(self, f'b{res}')(x, img, **block_kwargs)
        cmap = self.mapping(None, c) if (self.c_dim is not None and self.c_dim > 0) else None
        return self.b4(x, img, cmap)


# ----------------------------------------------------------------------------------------------------------------
# DCGAN (plumbing config: stock torch.nn layers, CPU eager -- reference :471-513)

class Discriminator_dcgan(torch.nn.Module):
    """four stride-2 5x5 convolutions (LeakyReLU, BatchNorm after all but the first) then a linear logit"""

    def __init__(self, M):
        super().__init__()
        layers = []
        for i, (cin, cout) in enumerate([(3, 64), (64, 128), (128, 256), (256, 512)]):
            layers += [torch.nn.Conv2d(cin, cout, 5, 2, 2, bias=False), torch.nn.LeakyReLU(0.2, inplace=True)]
            if i > 0:
                layers.append(torch.nn.BatchNorm2d(cout))
        self.main = torch.nn.Sequential(*layers)
        self.linear = torch.nn.Linear(M // 16 * M // 16 * 512, 1)

    def forward(self, x, c):
        return self.linear(torch.flatten(self.main(x), start_dim=1))


@discriminators.add_to_registry("cnn32_dcgan")
class Discriminator32_dcgan(Discriminator_dcgan):
    def __init__(self, *args, **kwargs):
        super().__init__(M=32)


@discriminators.add_to_registry("cnn48_dcgan")
class Discriminator48_dcgan(Discriminator_dcgan):
    def __init__(self):
        super().__init__(M=48)


# ----------------------------------------------------------------------------------------------------------------
# BigGAN discriminator (reference :699-842): spectral-norm residual blocks, self-attention, projection head

def D_arch(ch=64, attention='64', ksize='333333', dilation='111111'):
    att = [int(item) for item in attention.split('_')]
    arch = {}
    arch[256] = {'in_channels': [3] + [ch * m for m in [1, 2, 4, 8, 8, 16]], 'out_channels': [ch * m for m in [1, 2, 4, 8, 8, 16, 16]],
                 'downsample': [True] * 6 + [False], 'resolution': [128, 64, 32, 16, 8, 4, 4]}
    arch[128] = {'in_channels': [3] + [ch * m for m in [1, 2, 4, 8, 16]], 'out_channels': [ch * m for m in [1, 2, 4, 8, 16, 16]],
                 'downsample': [True] * 5 + [False], 'resolution': [64, 32, 16, 8, 4, 4]}
    arch[64] = {'in_channels': [3] + [ch * m for m in [1, 2, 4, 8]], 'out_channels': [ch * m for m in [1, 2, 4, 8, 16]],
                'downsample': [True] * 4 + [False], 'resolution': [32, 16, 8, 4, 4]}
    arch[32] = {'in_channels': [3] + [ch * m for m in [4, 4, 4]], 'out_channels': [ch * m for m in [4, 4, 4, 4]],
                'downsample': [True, True, False, False], 'resolution': [16, 16, 16, 16]}
    for res, a in arch.items():
        top = 8 if res > 32 else 6
        a['attention'] = {2 ** i: (2 ** i in att) for i in range(2, top)}
    return arch


@discriminators.add_to_registry("big_gan")
class BigGanDiscriminator(torch.nn.Module):
    def __init__(self, z_dim=128, c_dim=10, D_ch=64, D_wide=True, img_resolution=128,
                 D_kernel_size=3, D_attn='64', n_classes=10,
                 num_D_SVs=1, num_D_SV_itrs=1, D_activation='relu',
                 SN_eps=1e-12, output_dim=1, D_mixed_precision=False, D_fp16=False,
                 D_init='ortho', D_param='SN', **kwargs):
        super().__init__()
        import functools
        from ..biggan import layers
        self.z_dim, self.c_dim, self.ch, self.D_wide = z_dim, c_dim, D_ch, D_wide
        self.img_resolution, self.kernel_size, self.attention, self.n_classes = img_resolution, D_kernel_size, D_attn, n_classes
        assert D_activation == 'relu'
        self.activation = layers.ReLU()
        self.init, self.D_param, self.SN_eps, self.fp16 = D_init, D_param, SN_eps, D_fp16
        self.arch = D_arch(self.ch, self.attention)[img_resolution]
        assert self.D_param == 'SN'
        sn = dict(num_svs=num_D_SVs, num_itrs=num_D_SV_itrs, eps=self.SN_eps)
        self.which_conv = functools.partial(layers.SNConv2d, kernel_size=3, padding=1, **sn)
        self.which_linear = functools.partial(layers.SNLinear, **sn)
        self.which_embedding = functools.partial(layers.SNEmbedding, **sn)
        blocks = []
        for index in range(len(self.arch['out_channels'])):
            stage = [layers.DBlock(in_channels=self.arch['in_channels'][index], out_channels=self.arch['out_channels'][index],
                                   which_conv=self.which_conv, wide=self.D_wide, activation=self.activation, preactivation=(index > 0),
                                   downsample=(layers.avg_pool2x if self.arch['downsample'][index] else None))]
            if self.arch['attention'][self.arch['resolution'][index]]:
                stage.append(layers.Attention(self.arch['out_channels'][index], self.which_conv))
            blocks.append(torch.nn.ModuleList(stage))
        self.blocks = torch.nn.ModuleList(blocks)
        self.linear = self.which_linear(self.arch['out_channels'][-1], output_dim)
        self.embed = self.which_embedding(self.n_classes, self.arch['out_channels'][-1])
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

    def forward(self, x, c=None):
        from ..torch_utils.ops import modulate
        y = torch.argmax(c, dim=1) if c is not None else None
        h = x
        for blocklist in self.blocks:
            for block in blocklist:
                h = block(h)
        h = self.activation(h)
        h = modulate.dot_hw(h).to(h.dtype) if h.device.type == 'cuda' else torch.sum(h, [2, 3])      # global sum pooling
        out = self.linear(h)
        return out + torch.sum(self.embed(y) * h, 1, keepdim=True)
