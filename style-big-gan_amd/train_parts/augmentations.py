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


class _ColorTransform(torch.autograd.Function):
    """y = M[:, :, :3] @ x + M[:, :, 3:] per sample on planar RGB (one streaming kernel); linear in x, so the gradient is the
    same op with the transposed 3x3 block (`Mt`, zero offset) -- both matrices come from the host-side sampler."""

    @staticmethod
    def forward(ctx, x, M, Mt):
        _lib.require_cuda(x, "AugmentPipe")
        n, c, h, w = x.shape
        assert c == 3 and M.shape == (n, 3, 4) and Mt.shape == (n, 3, 4)
        xc = x.to(torch.float32).contiguous()
        y = torch.empty_like(xc)
        Mc = M.contiguous()
        if y.numel():
            _lib.check(_lib.load().sbg_color_transform(xc.data_ptr(), Mc.data_ptr(), y.data_ptr(), n, h * w, _lib.stream_ptr(x.device)), "sbg_color_transform")
        ctx.save_for_backward(M, Mt)
        return y.to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        M, Mt = ctx.saved_tensors
        zero_off = torch.cat([M[:, :, :3], torch.zeros_like(M[:, :, 3:])], dim=2)
        return (_ColorTransform.apply(dy, Mt, zero_off) if ctx.needs_input_grad[0] else None), None, None


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
        """Host copy of `p`, read (one synchronising read) whenever the buffer was written by someone who did not announce it --
        construction, ``load_state_dict``, user code.  An owner that updates `p` on the device every few iterations (StepEngine's ADA
        heuristic) announces the write instead and picks the point at which the new value takes effect, see below."""
        key = (self.p.data_ptr(), self.p._version)
        if self._p_cache[:2] != key:
            self._p_pending = None
            self._p_cache = key + (float(self.p),)
        return self._p_cache[2]

    def announce_strength_update(self):
        """`p` has just been rewritten on the device by the caller: start an asynchronous copy of the new value into pinned memory.
        The sampler keeps using the previous strength until ``adopt_strength()``."""
        value = self._strength() if self._p_cache[0] is None else self._p_cache[2]
        if self.p.device.type != "cuda":
            self._p_cache = (self.p.data_ptr(), self.p._version, float(self.p))
            return
        pinned = torch.empty([], dtype=self.p.dtype, pin_memory=True)
        pinned.copy_(self.p.detach(), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.p.device))
        self._p_pending = (pinned, ev)
        self._p_cache = (self.p.data_ptr(), self.p._version, value)

    def adopt_strength(self):
        """Make the announced value current.  Called at a point fixed in the iteration count (one iteration after the update), so the
        strength every sampler call sees is a function of the iteration alone -- the same on all ranks and from run to run; by then
        the copy has completed and the event wait returns at once."""
        pend = getattr(self, "_p_pending", None)
        if pend is not None:
            pend[1].synchronize()
            self._p_cache = self._p_cache[:2] + (float(pend[0]),)
            self._p_pending = None

    # -- parameters --------------------------------------------------------------------------------------------------
    def sample(self, batch_size, num_channels, height, width, debug_percentile=None, p=None):
        """Draw and compose the parameters of one call on the host.  Order and shapes of the random draws follow the reference's
        forward (:200-431) -- torch's CPU generator, so a CPU-seeded call reproduces the reference's CPU run -- and the arithmetic is
        float32 numpy (a few hundred microseconds per call).  Returns a dict of CPU tensors / ints: theta [N,2,3] + margins +
        up-sampled shape (geometry), color [N,3,4] (+ color_t for the gradient kernel), taps [N,T], sigma [N], cut [N,4];
        absent keys = stage disabled."""
        n, W, H = batch_size, width, height
        f32 = np.float32
        p = f32(self._strength() if p is None else p)
        dp = None if debug_percentile is None else torch.as_tensor(debug_percentile, dtype=torch.float32)
        rand = lambda *s: torch.rand(list(s)).numpy()
        randn = lambda *s: torch.randn(list(s)).numpy()
        erf = (lambda std: f32(float(torch.erfinv(dp * 2 - 1)) * std)) if dp is not None else None
        dpf = None if dp is None else f32(float(dp))
        out = dict()
        zeros, ones = np.zeros([n], f32), np.ones([n], f32)

        def gate(u, mult, value, off):
            return np.where(u < f32(mult) * p, value, off).astype(f32)

        def m3(a, b, c, d, e, f):           # [[a, b, c], [d, e, f], [0, 0, 1]] per sample
            m = np.zeros([n, 3, 3], f32)
            m[:, 0, 0], m[:, 0, 1], m[:, 0, 2], m[:, 1, 0], m[:, 1, 1], m[:, 1, 2], m[:, 2, 2] = a, b, c, d, e, f, 1
            return m

        def rot(th):
            return m3(np.cos(th), np.sin(-th), 0, np.sin(th), np.cos(th), 0)

        # pixel blitting + general geometric transforms: G_inv @ pixel_out ==> pixel_in
        G = None

        def chain(m):
            nonlocal G
            G = m if G is None else G @ m

        if self.xflip > 0:
            i = np.floor(rand(n) * 2)
            i = gate(rand(n), self.xflip, i, zeros)
            if dp is not None:
                i = np.full([n], np.floor(dpf * 2), f32)
            chain(m3(1 / (1 - 2 * i), 0, 0, 0, 1, 0))
        if self.rotate90 > 0:
            i = np.floor(rand(n) * 4)
            i = gate(rand(n), self.rotate90, i, zeros)
            if dp is not None:
                i = np.full([n], np.floor(dpf * 4), f32)
            chain(rot(f32(np.pi / 2) * i))
        if self.xint > 0:
            t = (rand(n, 2) * 2 - 1) * f32(self.xint_max)
            t = gate(rand(n, 1), self.xint, t, np.zeros([n, 2], f32))
            if dp is not None:
                t = np.full([n, 2], (dpf * 2 - 1) * f32(self.xint_max), f32)
            chain(m3(1, 0, -np.round(t[:, 0] * W), 0, 1, -np.round(t[:, 1] * H)))
        if self.scale > 0:
            s = np.exp2(randn(n) * f32(self.scale_std))
            s = gate(rand(n), self.scale, s, ones)
            if dp is not None:
                s = np.full([n], np.exp2(erf(self.scale_std)), f32)
            chain(m3(1 / s, 0, 0, 0, 1 / s, 0))
        p_rot = f32(1) - np.sqrt(np.clip(f32(1) - f32(self.rotate) * p, 0, 1))       # P(pre OR post) = p
        if self.rotate > 0:
            theta = (rand(n) * 2 - 1) * f32(np.pi) * f32(self.rotate_max)
            theta = np.where(rand(n) < p_rot, theta, zeros).astype(f32)
            if dp is not None:
                theta = np.full([n], (dpf * 2 - 1) * f32(np.pi) * f32(self.rotate_max), f32)
            chain(rot(theta))               # rotate2d_inv(-theta), before the anisotropic scaling
        if self.aniso > 0:
            s = np.exp2(randn(n) * f32(self.aniso_std))
            s = gate(rand(n), self.aniso, s, ones)
            if dp is not None:
                s = np.full([n], np.exp2(erf(self.aniso_std)), f32)
            chain(m3(1 / s, 0, 0, 0, 1 / (1 / s), 0))
        if self.rotate > 0:
            theta = (rand(n) * 2 - 1) * f32(np.pi) * f32(self.rotate_max)
            theta = np.where(rand(n) < p_rot, theta, zeros).astype(f32)
            if dp is not None:
                theta = zeros
            chain(rot(theta))               # after the anisotropic scaling
        if self.xfrac > 0:
            t = randn(n, 2) * f32(self.xfrac_std)
            t = gate(rand(n, 1), self.xfrac, t, np.zeros([n, 2], f32))
            if dp is not None:
                t = np.full([n, 2], erf(self.xfrac_std), f32)
            chain(m3(1, 0, -(t[:, 0] * W), 0, 1, -(t[:, 1] * H)))

        if G is not None:
            # padding that keeps every sampled position inside the (reflect-padded) image (:270-285)
            cx, cy = (W - 1) / 2, (H - 1) / 2
            cp = G @ np.asarray([[-cx, -cy, 1], [cx, -cy, 1], [cx, cy, 1], [-cx, cy, 1]], f32).T       # [N, xyz, corner]
            hz_pad = self.Hz_geom.shape[0] // 4
            margin = cp[:, :2, :].transpose(1, 0, 2).reshape(2, -1)             # [xy, N * corner]
            margin = np.concatenate([-margin, margin]).max(axis=1)              # [x0, y0, x1, y1]
            margin = margin + np.asarray([hz_pad * 2 - cx, hz_pad * 2 - cy] * 2, f32)
            margin = np.minimum(np.maximum(margin, 0), np.asarray([W - 1, H - 1] * 2, f32))
            mx0, my0, mx1, my1 = (int(v) for v in np.ceil(margin).astype(np.int32))

            def one(a, b, c, d, e, f):
                return np.asarray([[[a, b, c], [d, e, f], [0, 0, 1]]], f32)
            # origin shift of the padding, the 2x up-sampling, and normalisation to the [-1, 1] coordinates of the sampler (:288-300)
            G = one(1, 0, (mx0 - mx1) / 2, 0, 1, (my0 - my1) / 2) @ G
            G = one(2, 0, 0, 0, 2, 0) @ G @ one(1 / 2, 0, 0, 0, 1 / 2, 0)
            G = one(1, 0, -0.5, 0, 1, -0.5) @ G @ one(1, 0, 0.5, 0, 1, 0.5)
            up_h, up_w = (H + my0 + my1) * 2, (W + mx0 + mx1) * 2
            shape = [n, num_channels, (H + hz_pad * 2) * 2, (W + hz_pad * 2) * 2]
            G = one(2 / up_w, 0, 0, 0, 2 / up_h, 0) @ G @ one(1 / (2 / shape[3]), 0, 0, 0, 1 / (2 / shape[2]), 0)
            out.update(theta=torch.from_numpy(np.ascontiguousarray(G[:, :2, :], f32)), margins=(mx0, mx1, my0, my1), up_shape=(up_h, up_w),
                       grid_shape=shape, hz_pad=hz_pad)

        # colour transforms: C @ color_in ==> color_out
        C = None

        def cchain(m):
            nonlocal C
            C = m if C is None else m @ C

        def m4():
            m = np.zeros([n, 4, 4], f32)
            m[:, 0, 0] = m[:, 1, 1] = m[:, 2, 2] = m[:, 3, 3] = 1
            return m

        v = (np.asarray([1, 1, 1, 0]) / np.sqrt(3)).astype(f32)               # luma axis
        vv = np.outer(v, v).astype(f32)
        I4 = np.eye(4, dtype=f32)
        if self.brightness > 0:
            b = randn(n) * f32(self.brightness_std)
            b = gate(rand(n), self.brightness, b, zeros)
            if dp is not None:
                b = np.full([n], erf(self.brightness_std), f32)
            m = m4()
            m[:, 0, 3] = m[:, 1, 3] = m[:, 2, 3] = b
            cchain(m)
        if self.contrast > 0:
            c = np.exp2(randn(n) * f32(self.contrast_std))
            c = gate(rand(n), self.contrast, c, ones)
            if dp is not None:
                c = np.full([n], np.exp2(erf(self.contrast_std)), f32)
            m = m4()
            m[:, 0, 0] = m[:, 1, 1] = m[:, 2, 2] = c
            cchain(m)
        if self.lumaflip > 0:
            i = np.floor(rand(n, 1, 1) * 2)
            i = gate(rand(n, 1, 1), self.lumaflip, i, np.zeros([n, 1, 1], f32))
            if dp is not None:
                i = np.full([n, 1, 1], np.floor(dpf * 2), f32)
            cchain(I4 - 2 * vv * i)                                             # Householder reflection
        if self.hue > 0 and num_channels > 1:
            theta = (rand(n) * 2 - 1) * f32(np.pi) * f32(self.hue_max)
            theta = gate(rand(n), self.hue, theta, zeros)
            if dp is not None:
                theta = np.full([n], (dpf * 2 - 1) * f32(np.pi) * f32(self.hue_max), f32)
            vx, vy, vz = v[0], v[1], v[2]
            s, c = np.sin(theta), np.cos(theta)
            cc = 1 - c
            m = m4()
            m[:, 0, 0], m[:, 0, 1], m[:, 0, 2] = vx * vx * cc + c, vx * vy * cc - vz * s, vx * vz * cc + vy * s
            m[:, 1, 0], m[:, 1, 1], m[:, 1, 2] = vy * vx * cc + vz * s, vy * vy * cc + c, vy * vz * cc - vx * s
            m[:, 2, 0], m[:, 2, 1], m[:, 2, 2] = vz * vx * cc - vy * s, vz * vy * cc + vx * s, vz * vz * cc + c
            cchain(m)
        if self.saturation > 0 and num_channels > 1:
            s = np.exp2(randn(n, 1, 1) * f32(self.saturation_std))
            s = gate(rand(n, 1, 1), self.saturation, s, np.ones([n, 1, 1], f32))
            if dp is not None:
                s = np.full([n, 1, 1], np.exp2(erf(self.saturation_std)), f32)
            cchain(vv + (I4 - vv) * s)
        if C is not None:
            C = np.broadcast_to(C, [n, 4, 4])
            if num_channels == 3:
                out['color'] = torch.from_numpy(np.ascontiguousarray(C[:, :3, :], f32))
                ct = np.zeros([n, 3, 4], f32)
                ct[:, :, :3] = C[:, :3, :3].transpose(0, 2, 1)
                out['color_t'] = torch.from_numpy(ct)
            elif num_channels == 1:
                Cm = C[:, :3, :].mean(axis=1, keepdims=True)
                out['color'] = torch.from_numpy(np.concatenate([Cm[:, :, :3].sum(axis=2, keepdims=True), Cm[:, :, 3:]], axis=2).astype(f32))    # [N, 1, 2]: scale, offset
            else:
                raise ValueError('Image must be RGB (3 channels) or L (1 channel)')

        # image-space filter: per-sample gains of the four bands -> one separable filter per sample (:364-381)
        if self.imgfilter > 0:
            bank = self._fbank_host.numpy()
            nb = bank.shape[0]
            assert len(self.imgfilter_bands) == nb
            expected_power = (np.array([10, 1, 1, 1]) / 13).astype(f32)
            g = np.ones([n, nb], f32)
            for i, band_strength in enumerate(self.imgfilter_bands):
                t_i = np.exp2(randn(n) * f32(self.imgfilter_std))
                t_i = np.where(rand(n) < f32(self.imgfilter) * p * f32(band_strength), t_i, ones).astype(f32)
                if dp is not None:
                    t_i = np.full([n], np.exp2(erf(self.imgfilter_std)), f32) if band_strength > 0 else ones
                t = np.ones([n, nb], f32)
                t[:, i] = t_i
                t = t / np.sqrt((expected_power * np.square(t)).sum(axis=-1, keepdims=True))
                g = g * t
            out['taps'] = torch.from_numpy(np.ascontiguousarray(g @ bank, f32))          # [N, T]

        # corruptions
        if self.noise > 0:
            sigma = np.abs(randn(n, 1, 1, 1)) * f32(self.noise_std)
            sigma = gate(rand(n, 1, 1, 1), self.noise, sigma, np.zeros([n, 1, 1, 1], f32))
            if dp is not None:
                sigma = np.full([n, 1, 1, 1], f32(float(torch.erfinv(dp))) * f32(self.noise_std), f32)
            out['sigma'] = torch.from_numpy(sigma.reshape(n).astype(f32))
        if self.cutout > 0:
            size = np.full([n, 2, 1, 1, 1], self.cutout_size, f32)
            size = gate(rand(n, 1, 1, 1, 1), self.cutout, size, np.zeros_like(size))
            center = rand(n, 2, 1, 1, 1)
            if dp is not None:
                size = np.full([n, 2, 1, 1, 1], self.cutout_size, f32)
                center = np.full([n, 2, 1, 1, 1], dpf, f32)
            out['cut'] = torch.from_numpy(np.concatenate([center.reshape(n, 2), size.reshape(n, 2)], axis=1).astype(f32))      # cx, cy, sx, sy
        return out

    # -- image work --------------------------------------------------------------------------------------------------
    def apply(self, images, params, noise_image=None):
        """Run the transforms `params` (from sample()) describes on `images` [N, C, H, W] (device).  `noise_image`: the
        unit-variance noise field of the additive-noise stage (drawn on the device when None)."""
        _lib.require_cuda(images, "AugmentPipe")
        n, ch, H, W = images.shape
        device = images.device

        # one upload for every per-sample parameter
        names = [k for k in ('theta', 'color', 'color_t', 'taps', 'sigma', 'cut') if k in params]
        dev = dict()
        if names:
            flat = [params[k].reshape(params[k].shape[0], -1).expand(n, -1).to(torch.float32) for k in names]
            host = torch.cat(flat, dim=1)
            if device.type == "cuda":       # pinned staging buffer: the upload queues behind the running kernels instead of draining the stream
                host = host.pin_memory()
            packed = host.to(device, non_blocking=True)
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
                if 'color_t' in dev:
                    Ct = dev['color_t'].reshape(n, 3, 4)
                else:       # parameters that did not come from sample(): derive the gradient's matrix on the device
                    Ct = torch.cat([Cm[:, :, :3].transpose(1, 2), torch.zeros_like(Cm[:, :, 3:])], dim=2)
                images = _ColorTransform.apply(images, Cm, Ct)
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
