"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's ADA augmentation pipe.

Follows ``train_parts/augmentations.py:121-433`` of the reference (identical to ``stylegan2ada/training/augment.py``):
same random draws in the same order from torch's global CPU generator, same float32 arithmetic, stock
``F.affine_grid`` / ``F.grid_sample`` (what ``grid_sample_gradfix`` falls back to on CPU, grid_sample_gradfix.py:24-27) and the
oracle's own ``upfirdn2d``.  Pinned by ``tests/golden/augment.npz``, captured from the reference class under fixed seeds
(``tests/golden/make_golden.py``).  Imported by ``tests/`` only; the product never calls it.

``augment(images, cfg, p, debug_percentile=None, trace=None)``: ``cfg`` = the reference constructor's keyword arguments;
``trace`` (a dict) receives the composed per-sample parameters so that the HIP path can be run on exactly the same ones.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import ops

SYM2 = [-0.12940952255092145, 0.22414386804185735, 0.836516303737469, 0.48296291314469025]
SYM6 = [0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633, 0.4910559419267466,
        0.787641141030194, 0.3379294217276218, -0.07263752278646252, -0.021060292512300564, 0.04472490177066578,
        0.0017677118642428036, -0.007800708325034148]

DEFAULTS = dict(xflip=0, rotate90=0, xint=0, xint_max=0.125, scale=0, rotate=0, aniso=0, xfrac=0, scale_std=0.2, rotate_max=1,
                aniso_std=0.2, xfrac_std=0.125, brightness=0, contrast=0, lumaflip=0, hue=0, saturation=0, brightness_std=0.2,
                contrast_std=0.5, hue_max=1, saturation_std=1, imgfilter=0, imgfilter_bands=(1, 1, 1, 1), imgfilter_std=1,
                noise=0, cutout=0, noise_std=0.1, cutout_size=0.5)


def filter_bank():
    """reference :176-185 -- [4, taps] band-pass bank built from sym2"""
    lo = np.asarray(SYM2)
    hi = lo * ((-1) ** np.arange(lo.size))
    lo2, hi2 = np.convolve(lo, lo[::-1]) / 2, np.convolve(hi, hi[::-1]) / 2
    bank = np.eye(4, 1)
    for i in range(1, 4):
        stretched = np.zeros([4, bank.shape[1] * 2 - 1])
        stretched[:, ::2] = bank
        bank = np.stack([np.convolve(r, lo2) for r in stretched])
        c = bank.shape[1]
        bank[i, (c - hi2.size) // 2:(c + hi2.size) // 2] += hi2
    return torch.as_tensor(bank, dtype=torch.float32)


class _GridSampleFwd(torch.autograd.Function):
    """stock grid_sample made differentiable to any order w.r.t. the input -- what the reference's grid_sample_gradfix does on the
    torch versions it supports (grid_sample_gradfix.py:41-81): backward = aten::grid_sampler_2d_backward, whose own backward
    w.r.t. grad_output is the forward op again (linear in the input for a fixed grid)."""

    @staticmethod
    def forward(ctx, x, grid):
        ctx.save_for_backward(x, grid)
        return F.grid_sample(x, grid, mode='bilinear', padding_mode='zeros', align_corners=False)

    @staticmethod
    def backward(ctx, dy):
        x, grid = ctx.saved_tensors
        return _GridSampleBwd.apply(dy, x, grid), None


class _GridSampleBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, x, grid):
        ctx.save_for_backward(grid)
        dx, _ = torch.ops.aten.grid_sampler_2d_backward(dy, x, grid, 0, 0, False, [True, False])
        return dx

    @staticmethod
    def backward(ctx, g2):
        grid, = ctx.saved_tensors
        return _GridSampleFwd.apply(g2, grid), None, None


def grid_sample(x, grid):
    return _GridSampleFwd.apply(x, grid)


def _m(rows, like=None):
    """matrix from scalars / equally shaped tensors (reference `matrix`, :46-54)"""
    ts = [v for r in rows for v in r if isinstance(v, torch.Tensor)]
    if not ts:
        return torch.as_tensor(np.asarray(rows), dtype=torch.float32)
    shp = ts[0].shape
    el = [v if isinstance(v, torch.Tensor) else torch.full(shp, float(v)) for r in rows for v in r]
    return torch.stack(el, dim=-1).reshape(shp + (len(rows), -1))


def _t2(tx, ty):
    return _m([[1, 0, tx], [0, 1, ty], [0, 0, 1]])


def _s2(sx, sy):
    return _m([[sx, 0, 0], [0, sy, 0], [0, 0, 1]])


def _r2(th):
    return _m([[torch.cos(th), torch.sin(-th), 0], [torch.sin(th), torch.cos(th), 0], [0, 0, 1]])


def augment(images, cfg, p=1.0, debug_percentile=None, trace=None, noise_image=None):
    a = dict(DEFAULTS)
    a.update(cfg)
    N, CH, H, W = images.shape
    p = torch.as_tensor(p, dtype=torch.float32)
    dp = None if debug_percentile is None else torch.as_tensor(debug_percentile, dtype=torch.float32)
    tr = trace if trace is not None else dict()
    erf = lambda std: torch.erfinv(dp * 2 - 1) * std

    def pick(shape, prob, val, other):
        return torch.where(torch.rand(shape) < prob, val, other)

    # ---- blitting and geometry: G maps output pixels to input pixels (:196-262)
    G = None
    if a['xflip'] > 0:
        i = torch.floor(torch.rand([N]) * 2)
        i = pick([N], a['xflip'] * p, i, torch.zeros_like(i))
        if dp is not None:
            i = torch.full_like(i, torch.floor(dp * 2))
        m = _s2(1 / (1 - 2 * i), torch.ones_like(i))
        G = m if G is None else G @ m
    if a['rotate90'] > 0:
        i = torch.floor(torch.rand([N]) * 4)
        i = pick([N], a['rotate90'] * p, i, torch.zeros_like(i))
        if dp is not None:
            i = torch.full_like(i, torch.floor(dp * 4))
        m = _r2(-(-np.pi / 2 * i))
        G = m if G is None else G @ m
    if a['xint'] > 0:
        t = (torch.rand([N, 2]) * 2 - 1) * a['xint_max']
        t = pick([N, 1], a['xint'] * p, t, torch.zeros_like(t))
        if dp is not None:
            t = torch.full_like(t, (dp * 2 - 1) * a['xint_max'])
        m = _t2(-torch.round(t[:, 0] * W), -torch.round(t[:, 1] * H))
        G = m if G is None else G @ m
    if a['scale'] > 0:
        s = torch.exp2(torch.randn([N]) * a['scale_std'])
        s = pick([N], a['scale'] * p, s, torch.ones_like(s))
        if dp is not None:
            s = torch.full_like(s, torch.exp2(erf(a['scale_std'])))
        m = _s2(1 / s, 1 / s)
        G = m if G is None else G @ m
    p_rot = 1 - torch.sqrt((1 - a['rotate'] * p).clamp(0, 1))
    if a['rotate'] > 0:
        th = (torch.rand([N]) * 2 - 1) * np.pi * a['rotate_max']
        th = pick([N], p_rot, th, torch.zeros_like(th))
        if dp is not None:
            th = torch.full_like(th, (dp * 2 - 1) * np.pi * a['rotate_max'])
        m = _r2(-(-th))
        G = m if G is None else G @ m
    if a['aniso'] > 0:
        s = torch.exp2(torch.randn([N]) * a['aniso_std'])
        s = pick([N], a['aniso'] * p, s, torch.ones_like(s))
        if dp is not None:
            s = torch.full_like(s, torch.exp2(erf(a['aniso_std'])))
        m = _s2(1 / s, 1 / (1 / s))
        G = m if G is None else G @ m
    if a['rotate'] > 0:
        th = (torch.rand([N]) * 2 - 1) * np.pi * a['rotate_max']
        th = pick([N], p_rot, th, torch.zeros_like(th))
        if dp is not None:
            th = torch.zeros_like(th)
        m = _r2(-(-th))
        G = m if G is None else G @ m
    if a['xfrac'] > 0:
        t = torch.randn([N, 2]) * a['xfrac_std']
        t = pick([N, 1], a['xfrac'] * p, t, torch.zeros_like(t))
        if dp is not None:
            t = torch.full_like(t, erf(a['xfrac_std']))
        m = _t2(-(t[:, 0] * W), -(t[:, 1] * H))
        G = m if G is None else G @ m

    if G is not None:       # (:268-303)
        hz = torch.as_tensor(SYM6, dtype=torch.float32)
        hz = hz / hz.sum()                              # setup_filter(normalize=True); separable (12 taps >= 8)
        hp = hz.shape[0] // 4
        cx, cy = (W - 1) / 2, (H - 1) / 2
        corners = torch.as_tensor(np.asarray([[-cx, -cy, 1], [cx, -cy, 1], [cx, cy, 1], [-cx, cy, 1]]), dtype=torch.float32)
        cp = G @ corners.t()
        mg = cp[:, :2, :].permute(1, 0, 2).flatten(1)
        mg = torch.cat([-mg, mg]).max(dim=1).values
        mg = mg + torch.as_tensor([hp * 2 - cx, hp * 2 - cy] * 2, dtype=torch.float32)
        mg = mg.max(torch.zeros(4)).min(torch.as_tensor([W - 1, H - 1] * 2, dtype=torch.float32))
        mx0, my0, mx1, my1 = (int(v) for v in mg.ceil().to(torch.int32))
        images = F.pad(images, [mx0, mx1, my0, my1], mode='reflect')
        G = _t2((mx0 - mx1) / 2, (my0 - my1) / 2) @ G
        images = ops.upsample2d(images, hz, up=2)
        G = _s2(2, 2) @ G @ _s2(1 / 2, 1 / 2)
        G = _t2(-0.5, -0.5) @ G @ _t2(0.5, 0.5)
        shape = [N, CH, (H + hp * 2) * 2, (W + hp * 2) * 2]
        G = _s2(2 / images.shape[3], 2 / images.shape[2]) @ G @ _s2(1 / (2 / shape[3]), 1 / (2 / shape[2]))
        tr.update(theta=G[:, :2, :].clone(), margins=(mx0, mx1, my0, my1), up_shape=tuple(images.shape[2:]), grid_shape=shape, hz_pad=hp)
        grid = F.affine_grid(theta=G[:, :2, :], size=shape, align_corners=False)
        images = grid_sample(images, grid)
        images = ops.downsample2d(images, hz, down=2, padding=-hp * 2, flip_filter=True)

    # ---- colour: C maps input colours to output colours (:309-361)
    I4 = torch.eye(4)
    C = None
    v = torch.as_tensor(np.asarray([1, 1, 1, 0]) / np.sqrt(3), dtype=torch.float32)
    if a['brightness'] > 0:
        b = torch.randn([N]) * a['brightness_std']
        b = pick([N], a['brightness'] * p, b, torch.zeros_like(b))
        if dp is not None:
            b = torch.full_like(b, erf(a['brightness_std']))
        m = _m([[1, 0, 0, b], [0, 1, 0, b], [0, 0, 1, b], [0, 0, 0, 1]])
        C = m if C is None else m @ C
    if a['contrast'] > 0:
        c = torch.exp2(torch.randn([N]) * a['contrast_std'])
        c = pick([N], a['contrast'] * p, c, torch.ones_like(c))
        if dp is not None:
            c = torch.full_like(c, torch.exp2(erf(a['contrast_std'])))
        m = _m([[c, 0, 0, 0], [0, c, 0, 0], [0, 0, c, 0], [0, 0, 0, 1]])
        C = m if C is None else m @ C
    if a['lumaflip'] > 0:
        i = torch.floor(torch.rand([N, 1, 1]) * 2)
        i = pick([N, 1, 1], a['lumaflip'] * p, i, torch.zeros_like(i))
        if dp is not None:
            i = torch.full_like(i, torch.floor(dp * 2))
        m = I4 - 2 * v.ger(v) * i
        C = m if C is None else m @ C
    if a['hue'] > 0 and CH > 1:
        th = (torch.rand([N]) * 2 - 1) * np.pi * a['hue_max']
        th = pick([N], a['hue'] * p, th, torch.zeros_like(th))
        if dp is not None:
            th = torch.full_like(th, (dp * 2 - 1) * np.pi * a['hue_max'])
        x, y, z = v[0], v[1], v[2]
        s, c = torch.sin(th), torch.cos(th)
        k = 1 - c
        m = _m([[x * x * k + c, x * y * k - z * s, x * z * k + y * s, 0],
                [y * x * k + z * s, y * y * k + c, y * z * k - x * s, 0],
                [z * x * k - y * s, z * y * k + x * s, z * z * k + c, 0],
                [0, 0, 0, 1]])
        C = m if C is None else m @ C
    if a['saturation'] > 0 and CH > 1:
        s = torch.exp2(torch.randn([N, 1, 1]) * a['saturation_std'])
        s = pick([N, 1, 1], a['saturation'] * p, s, torch.ones_like(s))
        if dp is not None:
            s = torch.full_like(s, torch.exp2(erf(a['saturation_std'])))
        m = v.ger(v) + (I4 - v.ger(v)) * s
        C = m if C is None else m @ C
    if C is not None:
        flat = images.reshape([N, CH, H * W])
        if CH == 3:
            tr['color'] = C[:, :3, :].expand(N, 3, 4).clone()
            flat = C[:, :3, :3] @ flat + C[:, :3, 3:]
        elif CH == 1:
            Cm = C[:, :3, :].mean(dim=1, keepdims=True)
            tr['color'] = torch.cat([Cm[:, :, :3].sum(dim=2, keepdims=True), Cm[:, :, 3:]], dim=2).expand(N, 1, 2).clone()
            flat = flat * Cm[:, :, :3].sum(dim=2, keepdims=True) + Cm[:, :, 3:]
        else:
            raise ValueError('Image must be RGB (3 channels) or L (1 channel)')
        images = flat.reshape([N, CH, H, W])

    # ---- image-space filter (:367-391)
    if a['imgfilter'] > 0:
        bank = filter_bank()
        nb = bank.shape[0]
        power = torch.as_tensor(np.array([10, 1, 1, 1]) / 13, dtype=torch.float32)
        g = torch.ones([N, nb])
        for i, bs in enumerate(a['imgfilter_bands']):
            ti = torch.exp2(torch.randn([N]) * a['imgfilter_std'])
            ti = pick([N], a['imgfilter'] * p * bs, ti, torch.ones_like(ti))
            if dp is not None:
                ti = torch.full_like(ti, torch.exp2(erf(a['imgfilter_std']))) if bs > 0 else torch.ones_like(ti)
            t = torch.ones([N, nb])
            t[:, i] = ti
            t = t / (power * t.square()).sum(dim=-1, keepdims=True).sqrt()
            g = g * t
        taps = g @ bank
        tr['taps'] = taps.clone()
        k = taps.unsqueeze(1).repeat([1, CH, 1]).reshape([N * CH, 1, -1])
        q = bank.shape[1] // 2
        x = F.pad(images.reshape([1, N * CH, H, W]), [q, q, q, q], mode='reflect')
        x = F.conv2d(x, k.unsqueeze(2), groups=N * CH)
        x = F.conv2d(x, k.unsqueeze(3), groups=N * CH)
        images = x.reshape([N, CH, H, W])

    # ---- corruptions (:397-431)
    if a['noise'] > 0:
        sigma = torch.randn([N, 1, 1, 1]).abs() * a['noise_std']
        sigma = pick([N, 1, 1, 1], a['noise'] * p, sigma, torch.zeros_like(sigma))
        if dp is not None:
            sigma = torch.full_like(sigma, torch.erfinv(dp) * a['noise_std'])
        field = torch.randn([N, CH, H, W])
        if noise_image is not None:
            field = noise_image
        tr.update(sigma=sigma.reshape(N).clone(), noise_image=field)
        images = images + field * sigma
    if a['cutout'] > 0:
        size = torch.full([N, 2, 1, 1, 1], float(a['cutout_size']))
        size = pick([N, 1, 1, 1, 1], a['cutout'] * p, size, torch.zeros_like(size))
        center = torch.rand([N, 2, 1, 1, 1])
        if dp is not None:
            size = torch.full_like(size, float(a['cutout_size']))
            center = torch.full_like(center, dp)
        tr['cut'] = torch.cat([center.reshape(N, 2), size.reshape(N, 2)], dim=1)
        xs = torch.arange(W).reshape([1, 1, 1, -1])
        ys = torch.arange(H).reshape([1, 1, -1, 1])
        mx = ((xs + 0.5) / W - center[:, 0]).abs() >= size[:, 0] / 2
        my = ((ys + 0.5) / H - center[:, 1]).abs() >= size[:, 1] / 2
        images = images * torch.logical_or(mx, my).to(torch.float32)
    return images
