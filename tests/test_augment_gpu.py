"""GPU: the ADA augmentation pipe's HIP ops and the pipe itself against the CPU oracle / the reference's golden vectors.

Tolerances (fp32 everywhere): 1e-5 of the tensor's max magnitude for the ops on identical sampling positions; 2e-4 where the
sampling positions are generated inside the kernel from theta (affine_grid's float32 rounding differs from the kernel's by
~1e-7 in normalised coordinates, i.e. ~1e-4 pixel on the up-sampled image, times the local image gradient)."""
import pytest
import torch
import torch.nn.functional as F

import style_big_gan_amd
from golden_util import Golden, max_rel
from oracle import augment as OA
from style_big_gan_amd.torch_utils.ops import grid_sample_gradfix
from style_big_gan_amd.train_parts import augmentations as A

pytestmark = pytest.mark.gpu


def test_grid_sample_matches_torch_cpu(dev):
    torch.manual_seed(0)
    for (n, c, ih, iw, oh, ow) in [(2, 3, 9, 11, 7, 13), (1, 1, 16, 16, 16, 16), (3, 3, 40, 32, 64, 48)]:
        x = torch.randn(n, c, ih, iw)
        grid = torch.rand(n, oh, ow, 2) * 2.6 - 1.3          # a good part of the positions falls outside: zero padding
        dy = torch.randn(n, c, oh, ow)
        v = torch.randn(n, c, ih, iw)
        xr, gr, dyr = x.clone().requires_grad_(True), grid.clone().requires_grad_(True), dy.clone().requires_grad_(True)
        yr = F.grid_sample(xr, gr, mode='bilinear', padding_mode='zeros', align_corners=False)
        dxr, dgr = torch.autograd.grad((yr * dyr).sum(), [xr, gr])
        ddyr = F.grid_sample(v, grid, mode='bilinear', padding_mode='zeros', align_corners=False)     # d(dx . v)/d(dy): the op is linear in x

        xg, gg, dyg = x.to(dev).requires_grad_(True), grid.to(dev).requires_grad_(True), dy.to(dev).requires_grad_(True)
        yg = grid_sample_gradfix.grid_sample(xg, gg)
        dxg, dgg = torch.autograd.grad((yg * dyg).sum(), [xg, gg], create_graph=True)
        ddyg, = torch.autograd.grad((dxg * v.to(dev)).sum(), dyg)
        assert max_rel(yg, yr) < 1e-5 and max_rel(dxg, dxr) < 1e-5 and max_rel(dgg, dgr) < 1e-4 and max_rel(ddyg, ddyr) < 1e-5
    # empty batch
    assert grid_sample_gradfix.grid_sample(torch.zeros(0, 3, 4, 4, device=dev), torch.zeros(0, 5, 5, 2, device=dev)).shape == (0, 3, 5, 5)
    with pytest.raises(RuntimeError, match="no CPU path"):
        grid_sample_gradfix.grid_sample(torch.zeros(1, 3, 4, 4), torch.zeros(1, 5, 5, 2))


def test_affine_grid_sample_matches_torch_cpu(dev):
    torch.manual_seed(1)
    n, c, ih, iw, oh, ow = 4, 3, 70, 64, 44, 52
    x = torch.randn(n, c, ih, iw)
    theta = torch.eye(2, 3).repeat(n, 1, 1) + 0.3 * torch.randn(n, 2, 3)
    dy = torch.randn(n, c, oh, ow)
    xr = x.clone().requires_grad_(True)
    yr = F.grid_sample(xr, F.affine_grid(theta, [n, c, oh, ow], align_corners=False), mode='bilinear', padding_mode='zeros', align_corners=False)
    dxr, = torch.autograd.grad((yr * dy).sum(), xr)
    xg = x.to(dev).requires_grad_(True)
    yg = grid_sample_gradfix.affine_grid_sample(xg, theta.to(dev), [n, c, oh, ow])
    dxg, = torch.autograd.grad((yg * dy.to(dev)).sum(), xg)
    assert max_rel(yg, yr) < 2e-4 and max_rel(dxg, dxr) < 2e-4
    # non-contiguous input view and strided output consumer
    xs = torch.randn(n, c, ih, iw * 2, device=dev)[:, :, :, ::2]
    ys = grid_sample_gradfix.affine_grid_sample(xs, theta.to(dev), [n, c, oh, ow])
    assert max_rel(ys, grid_sample_gradfix.affine_grid_sample(xs.contiguous(), theta.to(dev), [n, c, oh, ow])) < 1e-6


def test_per_sample_filter_matches_grouped_conv(dev):
    torch.manual_seed(2)
    n, c, h, w, t = 3, 3, 20, 27, 7
    x = torch.randn(n, c, h, w)
    taps = torch.randn(n, t)
    k = taps.unsqueeze(1).repeat([1, c, 1]).reshape(n * c, 1, -1)
    for axis in (0, 1):
        xr = x.clone().requires_grad_(True)
        wgt = k.unsqueeze(2) if axis == 0 else k.unsqueeze(3)
        yr = F.conv2d(xr.reshape(1, n * c, h, w), wgt, groups=n * c).reshape(n, c, h - (t - 1) * (axis == 1), w - (t - 1) * (axis == 0))
        dy = torch.randn_like(yr).requires_grad_(True)
        dxr, = torch.autograd.grad((yr * dy).sum(), xr, create_graph=True)
        v = torch.randn_like(x)
        ddyr, = torch.autograd.grad((dxr * v).sum(), dy)
        xg, dyg = x.to(dev).requires_grad_(True), dy.detach().to(dev).requires_grad_(True)
        yg = A._Filter1d.apply(xg, taps.to(dev), axis, 0, False)
        dxg, = torch.autograd.grad((yg * dyg).sum(), xg, create_graph=True)
        ddyg, = torch.autograd.grad((dxg * v.to(dev)).sum(), dyg)
        assert max_rel(yg, yr) < 1e-5 and max_rel(dxg, dxr) < 1e-5 and max_rel(ddyg, ddyr) < 1e-5


def _run_case(g, case, dev, second_order=False):
    i = case["idx"]
    x = g.t("x/" + case["input"])
    trace = dict()
    torch.manual_seed(case["seed"])
    xo = x.clone().requires_grad_(True)
    yo = OA.augment(xo, case["kwargs"], p=case["p"], debug_percentile=case["debug_percentile"], trace=trace)
    pipe = A.AugmentPipe(**case["kwargs"]).to(dev)
    noise = trace.pop("noise_image", None)
    xg = x.to(dev).requires_grad_(True)
    yg = pipe.apply(xg, trace, noise_image=None if noise is None else noise.to(dev))
    return xo, yo, xg, yg


def test_pipe_matches_reference_golden(dev):
    """apply() on the parameters the oracle traced vs the REFERENCE's outputs and input gradients (tests/golden/augment.npz)"""
    g = Golden("augment")
    for case in g.meta["cases"]:
        i = case["idx"]
        xo, yo, xg, yg = _run_case(g, case, dev)
        geometric = any(case["kwargs"].get(k, 0) for k in ("xflip", "rotate90", "xint", "scale", "rotate", "aniso", "xfrac"))
        tol = 2e-4 if geometric else 1e-5
        assert yg.shape == g.t(f"y/{i}").shape
        assert max_rel(yg, g.t(f"y/{i}")) < tol, case
        dxg, = torch.autograd.grad((yg * g.t(f"w/{i}").to(dev)).sum(), xg)
        assert max_rel(dxg, g.t(f"dx/{i}")) < tol, case


def test_pipe_second_order_matches_oracle(dev):
    """R1 differentiates the discriminator's input gradient through the pipe: d/dw of |d(sum(y * w(y))) / dx|^2 style chain"""
    g = Golden("augment")
    for case in g.meta["cases"]:
        if case["spec"] not in ("bgc", "bgcfnc") or case["input"] != "rgb" or case["debug_percentile"] is not None:
            continue
        xo, yo, xg, yg = _run_case(g, case, dev)
        torch.manual_seed(77)
        q = torch.randn(yo.shape).requires_grad_(True)
        qg = q.detach().to(dev).requires_grad_(True)
        outs = []
        for x, y, qq in ((xo, yo, q), (xg, yg, qg)):
            d, = torch.autograd.grad((y * y * qq).sum(), x, create_graph=True)      # nonlinear head so the second derivative is not trivial
            r1 = d.square().sum()
            gq, gx = torch.autograd.grad(r1, [qq, x])
            outs.append((r1, gq, gx))
        assert abs(float(outs[1][0]) - float(outs[0][0])) < 1e-3 * abs(float(outs[0][0]))
        assert max_rel(outs[1][1], outs[0][1]) < 1e-3 and max_rel(outs[1][2], outs[0][2]) < 1e-3, case


def test_pipe_forward_at_benchmark_shape(dev):
    torch.manual_seed(3)
    pipe = A.augmentations["sg2_ada"](**A.augpipe_specs["bgc"]).to(dev)
    x = (torch.rand(8, 3, 256, 256, device=dev) * 2 - 1).requires_grad_(True)
    pipe.p.copy_(torch.as_tensor(0.6))
    y = pipe(x)
    assert y.shape == x.shape and torch.isfinite(y).all()
    y.square().sum().backward()
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().max()) > 0
    pipe.p.copy_(torch.as_tensor(0.0))          # strength 0: every transform is the identity; the orthogonal sym6 round trip reconstructs
    y0 = pipe(x.detach())
    assert max_rel(y0, x) < 2e-2


def test_training_step_with_ada(dev):
    """StepEngine with the pipe in front of D and the ADA heuristic running (reference trainers.py:575-584, 768-771)"""
    from style_big_gan_amd.torch_utils import training_stats
    from style_big_gan_amd.train_parts import trainers
    gk = dict(z_dim=64, c_dim=0, w_dim=64, img_resolution=32, img_channels=3, mapping_kwargs=dict(num_layers=2),
              synthesis_kwargs=dict(channel_base=1024, channel_max=64, num_fp16_res=2, block_kwargs=dict(conv_clamp=256)))
    dk = dict(c_dim=0, img_resolution=32, img_channels=3, architecture='orig', channel_base=1024, channel_max=64, num_fp16_res=2,
              conv_clamp=256, epilogue_kwargs=dict(mbstd_group_size=4))
    training_stats.init_multiprocessing(rank=0, sync_device=None)
    eng = trainers.StepEngine(dev, gen_kwargs=gk, disc_kwargs=dk, loss_arch_kwargs=dict(style_mixing_prob=0), dis_regs=[('r1', dict(r1_gamma=0.01))],
                              d_reg_interval=2, batch=8, batch_gpu=4, augment_kwargs=dict(A.augpipe_specs['bgc']), augment_p=0.2,
                              ada_target=0.6, ada_interval=2, ada_kimg=0.1)
    assert eng.augment_pipe is not None and eng.loss.augment_pipe is eng.augment_pipe and abs(float(eng.augment_pipe.p) - 0.2) < 1e-6
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    strengths = []
    for _ in range(4):
        real = torch.rand([8, 3, 32, 32], device=dev, generator=gen) * 2 - 1
        eng.train_iteration(real, None)
        strengths.append(float(eng.augment_pipe.p))
    step = 8 * 2 / (0.1 * 1000)
    assert strengths[0] == pytest.approx(0.2)                                       # no adjustment before `ada_interval` iterations
    assert abs(abs(strengths[1] - 0.2) - step) < 1e-5                               # then one step of batch * interval / (ada_kimg * 1000), up or down
    assert abs(abs(strengths[3] - strengths[1]) - step) < 1e-5 or strengths[3] == 0.0
    assert all(torch.isfinite(p).all() for p in list(eng.G.parameters()) + list(eng.D.parameters()))
