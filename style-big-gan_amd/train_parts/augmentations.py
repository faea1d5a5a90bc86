"""ADA augmentation pipe ("Training Generative Adversarial Networks with Limited Data"), MI355X host side.

Counterpart of the reference's ``train_parts/augmentations.py`` (:121-433; the same class as
``stylegan2ada/training/augment.py``): registry name ``sg2_ada``, the same constructor arguments, the buffers ``p``,
``Hz_geom`` and ``Hz_fbank``, ``forward(images, debug_percentile=None)``, applied in front of every discriminator call
(``losses_base.py:44-45``) and differentiable to second order with respect to the images (R1 runs through it).

What is different, and why:

* The reference builds every per-sample parameter with device-side micro-ops (about 300 launches of a few numbers each)
  and then synchronises the host to read the padding margins (``margin.ceil().to(torch.int32)`` used as Python ints, :282-285).
  Here ``sample()`` draws and composes ALL parameters on the host (float32 CPU tensors, a few dozen microseconds), in the
  reference's order of random draws, and uploads them as ONE packed tensor: no device micro-ops, no synchronisation.  The
  strength ``p`` stays a device buffer for state-dict / ADA-heuristic parity; its host mirror is refreshed only when the
  buffer's version counter changes (every ``ada_interval`` iterations).
* ``apply()`` runs the image work as a handful of HIP launches: reflect pad, 2x up-sampling with the 12-tap sym6 low-pass
  (``upfirdn2d``), ``affine_grid_sample`` (affine grid generated inside the sampling kernel, no ``[N,H,W,2]`` grid in HBM),
  2x down-sampling; one batched 3x4 colour transform; per-sample separable band filter (``sbg_filter1d_batch``); noise; cutout.

The split also gives the parity tests their handle: the CPU oracle traces the parameters it drew, ``apply()`` is held to
the oracle on identical parameters, and ``sample()`` is held to the oracle's trace under the same CPU seed.
"""
import numpy as np
import torch

from .. import utils
from .. import _lib
from ..torch_utils.ops import upfirdn2d
from ..torch_utils.ops import grid_sample_gradfix

augmentations = utils.ClassRegistry()

# low-pass decomposition filters of the two orthogonal wavelets the pipe uses (values as published for PyWavelets'
# 'sym6' / 'sym2'; reference table: augmentations.py:24-41)
wavelets = {
    'sym2': [-0.12940952255092145, 0.22414386804185735, 0.836516303737469, 0.48296291314469025],
    'sym6': [0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633, 0.4910559419267466,
             0.787641141030194, 0.3379294217276218, -0.07263752278646252, -0.021060292512300564, 0.04472490177066578,
             0.0017677118642428036, -0.007800708325034148],
}

# named subsets of the augmentations (stylegan2ada/train.py:271-283); `config.aug.augpipe` selects one
augpipe_specs = {
    'blit':   dict(xflip=1, rotate90=1, xint=1),
    'geom':   dict(scale=1, rotate=1, aniso=1, xfrac=1),
    'color':  dict(brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1),
    'filter': dict(imgfilter=1),
    'noise':  dict(noise=1),
    'cutout': dict(cutout=1),
}
augpipe_specs['bg'] = {**augpipe_specs['blit'], **augpipe_specs['geom']}
augpipe_specs['bgc'] = {**augpipe_specs['bg'], **augpipe_specs['color']}
augpipe_specs['bgcf'] = {**augpipe_specs['bgc'], **augpipe_specs['filter']}
augpipe_specs['bgcfn'] = {**augpipe_specs['bgcf'], **augpipe_specs['noise']}
augpipe_specs['bgcfnc'] = {**augpipe_specs['bgcfn'], **augpipe_specs['cutout']}


# ----------------------------------------------------------------------------------------------------------------
# host-side homogeneous transforms, batched: every entry is a float or a [N] float32 CPU tensor

def _mat(rows, n):
    out = torch.zeros([n, len(rows), len(rows[0])], dtype=torch.float32)
    for i, row in enumerate(rows):
        for j, v in enumerate(row):
            out[:, i, j] = v
    return out


def _translate2d(tx, ty, n):
    return _mat([[1, 0, tx], [0, 1, ty], [0, 0, 1]], n)


def _scale2d(sx, sy, n):
    return _mat([[sx, 0, 0], [0, sy, 0], [0, 0, 1]], n)


def _rotate2d(theta, n):
    c, s = torch.cos(theta), torch.sin(theta)
    return _mat([[c, torch.sin(-theta), 0], [s, c, 0], [0, 0, 1]], n)


def _translate3d(tx, ty, tz, n):
    return _mat([[1, 0, 0, tx], [0, 1, 0, ty], [0, 0, 1, tz], [0, 0, 0, 1]], n)


def _scale3d(sx, sy, sz, n):
    return _mat([[sx, 0, 0, 0], [0, sy, 0, 0], [0, 0, sz, 0], [0, 0, 0, 1]], n)


def _rotate3d(v, theta, n):
    vx, vy, vz = float(v[0]), float(v[1]), float(v[2])
    s, c = torch.sin(theta), torch.cos(theta)
    cc = 1 - c
    return _mat([[vx * vx * cc + c, vx * vy * cc - vz * s, vx * vz * cc + vy * s, 0],
                 [vy * vx * cc + vz * s, vy * vy * cc + c, vy * vz * cc - vx * s, 0],
                 [vz * vx * cc - vy * s, vz * vy * cc + vx * s, vz * vz * cc + c, 0],
                 [0, 0, 0, 1]], n)


class _Filter1d(torch.autograd.Function):
    """per-sample correlation along W (axis 0) or H (axis 1) of [N, C, H, W] fp32 with taps [N, T]; linear, so its
    gradient is the same op with flipped taps and complementary zero padding (any order)."""

    @staticmethod
    def forward(ctx, x, taps, axis, pad, flip):
        _lib.require_cuda(x, "imgfilter")
        n, c, h, w = x.shape
        t = taps.shape[1]
        assert taps.shape[0] == n and taps.dtype == torch.float32 and not taps.requires_grad
        xc = x.to(torch.float32).contiguous()
        oh = h + 2 * pad - t + 1 if axis == 1 else h
        ow = w + 2 * pad - t + 1 if axis == 0 else w
        y = torch.empty([n, c, oh, ow], dtype=torch.float32, device=x.device)
        tc = taps.contiguous()
        if y.numel():
            _lib.check(_lib.load().sbg_filter1d_batch(xc.data_ptr(), tc.data_ptr(), y.data_ptr(), n * c, h, w, t, axis, pad, c, int(flip),
                                                      _lib.stream_ptr(x.device)), "sbg_filter1d_batch")
        ctx.save_for_backward(tc)
        ctx.cfg = (axis, pad, flip, t, x.dtype)
        return y.to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        tc, = ctx.saved_tensors
        axis, pad, flip, t, dtype = ctx.cfg
        dx = _Filter1d.apply(dy, tc, axis, t - 1 - pad, not flip) if ctx.needs_input_grad[0] else None
        return dx, None, None, None, None


@augmentations.add_to_registry("sg2_ada")
class AugmentPipe(torch.nn.Module):
    def __init__(self,
        xflip=0, rotate90=0, xint=0, xint_max=0.125,
        scale=0, rotate=0, aniso=0, xfrac=0, scale_std=0.2, rotate_max=1, aniso_std=0.2, xfrac_std=0.125,
        brightness=0, contrast=0, lumaflip=0, hue=0, saturation=0, brightness_std=0.2, contrast_std=0.5, hue_max=1, saturation_std=1,
        imgfilter=0, imgfilter_bands=(1, 1, 1, 1), imgfilter_std=1,
        noise=0, cutout=0, noise_std=0.1, cutout_size=0.5,
    ):
        super().__init__()
        self.register_buffer('p', torch.ones([]))       # overall multiplier of every augmentation probability (the ADA strength)
        for k, v in dict(xflip=xflip, rotate90=rotate90, xint=xint, xint_max=xint_max, scale=scale, rotate=rotate, aniso=aniso,
                         xfrac=xfrac, scale_std=scale_std, rotate_max=rotate_max, aniso_std=aniso_std, xfrac_std=xfrac_std,
                         brightness=brightness, contrast=contrast, lumaflip=lumaflip, hue=hue, saturation=saturation,
                         brightness_std=brightness_std, contrast_std=contrast_std, hue_max=hue_max, saturation_std=saturation_std,
                         imgfilter=imgfilter, imgfilter_std=imgfilter_std, noise=noise, cutout=cutout, noise_std=noise_std,
                         cutout_size=cutout_size).items():
            setattr(self, k, float(v))
        self.imgfilter_bands = list(imgfilter_bands)

        # orthogonal low-pass for the geometric transforms (reference :173)
        self.register_buffer('Hz_geom', upfirdn2d.setup_filter(wavelets['sym6']))

        # band-pass filter bank of the image-space filter (reference :176-185): row i = Bandpass(H(z), band i)
        lo = np.asarray(wavelets['sym2'])
        hi = lo * ((-1) ** np.arange(lo.size))
        lo2 = np.convolve(lo, lo[::-1]) / 2
        hi2 = np.convolve(hi, hi[::-1]) / 2
        fbank = np.eye(4, 1)
        for i in range(1, fbank.shape[0]):
            fbank = np.dstack([fbank, np.zeros_like(fbank)]).reshape(fbank.shape[0], -1)[:, :-1]     # zero-insert (z -> z^2)
            fbank = np.stack([np.convolve(row, lo2) for row in fbank])
            mid = fbank.shape[1]
            fbank[i, (mid - hi2.size) // 2:(mid + hi2.size) // 2] += hi2
        self.register_buffer('Hz_fbank', torch.as_tensor(fbank, dtype=torch.float32))
        self._p_cache = (None, None, 1.0)       # (data_ptr, version, value) of the host mirror of `p`
        self._fbank_host = torch.as_tensor(fbank, dtype=torch.float32)

    # -- strength mirror ---------------------------------------------------------------------------------------------
    def _strength(self):
        key = (self.p.data_ptr(), self.p._version)
        if self._p_cache[:2] != key:
            self._p_cache = key + (float(self.p),)      # the only host read of the pipe; happens when `p` was written
        return self._p_cache[2]

    # -- parameters --------------------------------------------------------------------------------------------------
    def sample(self, batch_size, num_channels, height, width, debug_percentile=None, p=None):
        """Draw and compose the parameters of one call on the host.  Order and shapes of the random draws follow the reference's
        forward (:200-431), so a CPU-seeded call reproduces the reference's CPU run.  Returns a dict of CPU tensors / ints:
        theta [N,2,3] + margins + up-sampled shape (geometry), color [N,3,4], taps [N,T], sigma [N], cut [N,4]; absent keys =
        stage disabled."""
        n, W, H = batch_size, width, height
        p = torch.as_tensor(self._strength() if p is None else p, dtype=torch.float32)
        dp = None if debug_percentile is None else torch.as_tensor(debug_percentile, dtype=torch.float32)
        rand, randn = torch.rand, torch.randn
        out = dict()

        def gate(shape, mult, value, off):
            return torch.where(rand(shape) < mult * p, value, off)

        # pixel blitting + general geometric transforms: G_inv @ pixel_out ==> pixel_in
        G = None

        def chain(m):
            nonlocal G
            G = m if G is None else G @ m

        if self.xflip > 0:
            i = torch.floor(rand([n]) * 2)
            i = gate([n], self.xflip, i, torch.zeros_like(i))
            if dp is not None:
                i = torch.full_like(i, torch.floor(dp * 2))
            chain(_scale2d(1 / (1 - 2 * i), 1, n))
        if self.rotate90 > 0:
            i = torch.floor(rand([n]) * 4)
            i = gate([n], self.rotate90, i, torch.zeros_like(i))
            if dp is not None:
                i = torch.full_like(i, torch.floor(dp * 4))
            chain(_rotate2d(np.pi / 2 * i, n))
        if self.xint > 0:
            t = (rand([n, 2]) * 2 - 1) * self.xint_max
            t = gate([n, 1], self.xint, t, torch.zeros_like(t))
            if dp is not None:
                t = torch.full_like(t, (dp * 2 - 1) * self.xint_max)
            chain(_translate2d(-torch.round(t[:, 0] * W), -torch.round(t[:, 1] * H), n))
        if self.scale > 0:
            s = torch.exp2(randn([n]) * self.scale_std)
            s = gate([n], self.scale, s, torch.ones_like(s))
            if dp is not None:
                s = torch.full_like(s, torch.exp2(torch.erfinv(dp * 2 - 1) * self.scale_std))
            chain(_scale2d(1 / s, 1 / s, n))
        p_rot = 1 - torch.sqrt((1 - self.rotate * p).clamp(0, 1))       # P(pre OR post) = p
        if self.rotate > 0:
            theta = (rand([n]) * 2 - 1) * np.pi * self.rotate_max
            theta = torch.where(rand([n]) < p_rot, theta, torch.zeros_like(theta))
            if dp is not None:
                theta = torch.full_like(theta, (dp * 2 - 1) * np.pi * self.rotate_max)
            chain(_rotate2d(theta, n))      # rotate2d_inv(-theta), before the anisotropic scaling
        if self.aniso > 0:
            s = torch.exp2(randn([n]) * self.aniso_std)
            s = gate([n], self.aniso, s, torch.ones_like(s))
            if dp is not None:
                s = torch.full_like(s, torch.exp2(torch.erfinv(dp * 2 - 1) * self.aniso_std))
            chain(_scale2d(1 / s, 1 / (1 / s), n))
        if self.rotate > 0:
            theta = (rand([n]) * 2 - 1) * np.pi * self.rotate_max
            theta = torch.where(rand([n]) < p_rot, theta, torch.zeros_like(theta))
            if dp is not None:
                theta = torch.zeros_like(theta)
            chain(_rotate2d(theta, n))      # after the anisotropic scaling
        if self.xfrac > 0:
            t = randn([n, 2]) * self.xfrac_std
            t = gate([n, 1], self.xfrac, t, torch.zeros_like(t))
            if dp is not None:
                t = torch.full_like(t, torch.erfinv(dp * 2 - 1) * self.xfrac_std)
            chain(_translate2d(-(t[:, 0] * W), -(t[:, 1] * H), n))

        if G is not None:
            # padding that keeps every sampled position inside the (reflect-padded) image (:270-285)
            cx, cy = (W - 1) / 2, (H - 1) / 2
            cp = torch.tensor([[-cx, -cy, 1], [cx, -cy, 1], [cx, cy, 1], [-cx, cy, 1]], dtype=torch.float32)
            cp = G @ cp.t()                                                 # [N, xyz, corner]
            hz_pad = self.Hz_geom.shape[0] // 4
            margin = cp[:, :2, :].permute(1, 0, 2).flatten(1)               # [xy, N * corner]
            margin = torch.cat([-margin, margin]).max(dim=1).values         # [x0, y0, x1, y1]
            margin = margin + torch.tensor([hz_pad * 2 - cx, hz_pad * 2 - cy] * 2, dtype=torch.float32)
            margin = margin.max(torch.zeros(4)).min(torch.tensor([W - 1, H - 1] * 2, dtype=torch.float32))
            mx0, my0, mx1, my1 = (int(v) for v in margin.ceil().to(torch.int32))
            # origin shift of the padding, the 2x up-sampling, and normalisation to the [-1, 1] coordinates of the sampler (:288-300)
            G = _translate2d((mx0 - mx1) / 2, (my0 - my1) / 2, 1) @ G
            G = _scale2d(2, 2, 1) @ G @ _scale2d(1 / 2, 1 / 2, 1)
            G = _translate2d(-0.5, -0.5, 1) @ G @ _translate2d(0.5, 0.5, 1)
            up_h, up_w = (H + my0 + my1) * 2, (W + mx0 + mx1) * 2
            shape = [n, num_channels, (H + hz_pad * 2) * 2, (W + hz_pad * 2) * 2]
            G = _scale2d(2 / up_w, 2 / up_h, 1) @ G @ _scale2d(1 / (2 / shape[3]), 1 / (2 / shape[2]), 1)
            out.update(theta=G[:, :2, :].contiguous(), margins=(mx0, mx1, my0, my1), up_shape=(up_h, up_w), grid_shape=shape, hz_pad=hz_pad)

        # colour transforms: C @ color_in ==> color_out
        C = None

        def cchain(m):
            nonlocal C
            C = m if C is None else m @ C

        v = torch.as_tensor(np.asarray([1, 1, 1, 0]) / np.sqrt(3), dtype=torch.float32)      # luma axis
        vv = v.ger(v)
        I4 = torch.eye(4)
        if self.brightness > 0:
            b = randn([n]) * self.brightness_std
            b = gate([n], self.brightness, b, torch.zeros_like(b))
            if dp is not None:
                b = torch.full_like(b, torch.erfinv(dp * 2 - 1) * self.brightness_std)
            cchain(_translate3d(b, b, b, n))
        if self.contrast > 0:
            c = torch.exp2(randn([n]) * self.contrast_std)
            c = gate([n], self.contrast, c, torch.ones_like(c))
            if dp is not None:
                c = torch.full_like(c, torch.exp2(torch.erfinv(dp * 2 - 1) * self.contrast_std))
            cchain(_scale3d(c, c, c, n))
        if self.lumaflip > 0:
            i = torch.floor(rand([n, 1, 1]) * 2)
            i = gate([n, 1, 1], self.lumaflip, i, torch.zeros_like(i))
            if dp is not None:
                i = torch.full_like(i, torch.floor(dp * 2))
            cchain(I4 - 2 * vv * i)                                         # Householder reflection
        if self.hue > 0 and num_channels > 1:
            theta = (rand([n]) * 2 - 1) * np.pi * self.hue_max
            theta = gate([n], self.hue, theta, torch.zeros_like(theta))
            if dp is not None:
                theta = torch.full_like(theta, (dp * 2 - 1) * np.pi * self.hue_max)
            cchain(_rotate3d(v, theta, n))
        if self.saturation > 0 and num_channels > 1:
            s = torch.exp2(randn([n, 1, 1]) * self.saturation_std)
            s = gate([n, 1, 1], self.saturation, s, torch.ones_like(s))
            if dp is not None:
                s = torch.full_like(s, torch.exp2(torch.erfinv(dp * 2 - 1) * self.saturation_std))
            cchain(vv + (I4 - vv) * s)
        if C is not None:
            if num_channels == 3:
                out['color'] = C[:, :3, :].expand(n, 3, 4).contiguous()
            elif num_channels == 1:
                Cm = C[:, :3, :].mean(dim=1, keepdims=True).expand(n, 1, 4)
                out['color'] = torch.cat([Cm[:, :, :3].sum(dim=2, keepdims=True), Cm[:, :, 3:]], dim=2).contiguous()    # [N, 1, 2]: scale, offset
            else:
                raise ValueError('Image must be RGB (3 channels) or L (1 channel)')

        # image-space filter: per-sample gains of the four bands -> one separable filter per sample (:364-381)
        if self.imgfilter > 0:
            nb = self._fbank_host.shape[0]
            assert len(self.imgfilter_bands) == nb
            expected_power = torch.as_tensor(np.array([10, 1, 1, 1]) / 13, dtype=torch.float32)
            g = torch.ones([n, nb])
            for i, band_strength in enumerate(self.imgfilter_bands):
                t_i = torch.exp2(randn([n]) * self.imgfilter_std)
                t_i = torch.where(rand([n]) < self.imgfilter * p * band_strength, t_i, torch.ones_like(t_i))
                if dp is not None:
                    t_i = torch.full_like(t_i, torch.exp2(torch.erfinv(dp * 2 - 1) * self.imgfilter_std)) if band_strength > 0 else torch.ones_like(t_i)
                t = torch.ones([n, nb])
                t[:, i] = t_i
                t = t / (expected_power * t.square()).sum(dim=-1, keepdims=True).sqrt()
                g = g * t
            out['taps'] = (g @ self._fbank_host).contiguous()              # [N, T]

        # corruptions
        if self.noise > 0:
            sigma = randn([n, 1, 1, 1]).abs() * self.noise_std
            sigma = gate([n, 1, 1, 1], self.noise, sigma, torch.zeros_like(sigma))
            if dp is not None:
                sigma = torch.full_like(sigma, torch.erfinv(dp) * self.noise_std)
            out['sigma'] = sigma.reshape(n)
        if self.cutout > 0:
            size = torch.full([n, 2, 1, 1, 1], self.cutout_size)
            size = gate([n, 1, 1, 1, 1], self.cutout, size, torch.zeros_like(size))
            center = rand([n, 2, 1, 1, 1])
            if dp is not None:
                size = torch.full_like(size, self.cutout_size)
                center = torch.full_like(center, dp)
            out['cut'] = torch.cat([center.reshape(n, 2), size.reshape(n, 2)], dim=1)      # cx, cy, sx, sy
        return out

    # -- image work --------------------------------------------------------------------------------------------------
    def apply(self, images, params, noise_image=None):
        """Run the transforms `params` (from sample()) describes on `images` [N, C, H, W] (device).  `noise_image`: the
        unit-variance noise field of the additive-noise stage (drawn on the device when None)."""
        _lib.require_cuda(images, "AugmentPipe")
        n, ch, H, W = images.shape
        device = images.device

        # one upload for every per-sample parameter
        names = [k for k in ('theta', 'color', 'taps', 'sigma', 'cut') if k in params]
        dev = dict()
        if names:
            flat = [params[k].reshape(params[k].shape[0], -1).expand(n, -1).to(torch.float32) for k in names]
            packed = torch.cat(flat, dim=1).to(device, non_blocking=True)
            off = 0
            for k, f in zip(names, flat):
                dev[k] = packed[:, off:off + f.shape[1]]
                off += f.shape[1]

        if 'theta' in params:
            mx0, mx1, my0, my1 = params['margins']
            images = torch.nn.functional.pad(input=images, pad=[mx0, mx1, my0, my1], mode='reflect')
            images = upfirdn2d.upsample2d(x=images, f=self.Hz_geom, up=2)
            assert tuple(images.shape[2:]) == tuple(params['up_shape'])
            theta = dev['theta'].reshape(n, 2, 3).contiguous()
            images = grid_sample_gradfix.affine_grid_sample(images, theta, params['grid_shape'],
                                                            theta_host=params['theta'].to(torch.float32).expand(n, 2, 3).contiguous())
            images = upfirdn2d.downsample2d(x=images, f=self.Hz_geom, down=2, padding=-params['hz_pad'] * 2, flip_filter=True)

        if 'color' in params:
            if ch == 3:
                Cm = dev['color'].reshape(n, 3, 4)
                images = torch.baddbmm(Cm[:, :, 3:], Cm[:, :, :3], images.reshape(n, 3, H * W)).reshape(n, 3, H, W)
            else:
                Cm = dev['color'].reshape(n, 1, 2, 1)
                images = images * Cm[:, :, :1] + Cm[:, :, 1:]

        if 'taps' in params:
            taps = dev['taps'].contiguous()
            pad = taps.shape[1] // 2
            images = torch.nn.functional.pad(input=images, pad=[pad, pad, pad, pad], mode='reflect')
            images = _Filter1d.apply(images, taps, 0, 0, False)
            images = _Filter1d.apply(images, taps, 1, 0, False)

        if 'sigma' in params:
            if noise_image is None:
                noise_image = torch.randn([n, ch, H, W], device=device)
            images = images + noise_image * dev['sigma'].reshape(n, 1, 1, 1)

        if 'cut' in params:
            cut = dev['cut']
            coord_x = torch.arange(W, device=device).reshape([1, 1, 1, -1])
            coord_y = torch.arange(H, device=device).reshape([1, 1, -1, 1])
            mask_x = (((coord_x + 0.5) / W - cut[:, 0].reshape(n, 1, 1, 1)).abs() >= cut[:, 2].reshape(n, 1, 1, 1) / 2)
            mask_y = (((coord_y + 0.5) / H - cut[:, 1].reshape(n, 1, 1, 1)).abs() >= cut[:, 3].reshape(n, 1, 1, 1) / 2)
            images = images * torch.logical_or(mask_x, mask_y).to(torch.float32)
        return images

    def forward(self, images, debug_percentile=None):
        assert isinstance(images, torch.Tensor) and images.ndim == 4
        n, ch, H, W = images.shape
        return self.apply(images, self.sample(n, ch, H, W, debug_percentile=debug_percentile))
