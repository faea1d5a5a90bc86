"""GPU parity: HIP ops (through the C ABI) vs the CPU oracle on the same seeded inputs.

Tolerances (stated per dtype): fp32 tensors -- elementwise ops 1e-5 abs/rel, MFMA convolutions (bf16x3 split) 2e-4
relative to the output's max magnitude; bf16 / f16 tensors -- 2e-2 relative to the output's max magnitude
(8-bit / 11-bit mantissa storage, fp32 accumulation).
"""
import itertools

import numpy as np
import pytest
import torch

import style_big_gan_amd  # noqa: F401
from style_big_gan_amd.torch_utils.ops import bias_act, conv2d_gradfix, conv2d_resample, fma, upfirdn2d
from oracle import ops as O

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def check(a, b, tol, what=""):
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    e = rel_err(a, b)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"


TOL = {torch.float32: 1e-5, torch.bfloat16: 2e-2, torch.float16: 4e-3}
CONV_TOL = {torch.float32: 2e-4, torch.bfloat16: 2e-2, torch.float16: 4e-3}


# ---------------------------------------------------------------------------------------------------- bias_act

@pytest.mark.parametrize("act", list(bias_act.activation_funcs.keys()))
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bias_act_forward_and_grads(dev, act, dtype):
    torch.manual_seed(1)
    for shape, dim, cl in [((2, 5, 4, 4), 1, False), ((3, 16, 6, 5), 1, True), ((3, 7), 1, False), ((4, 8, 3, 3), 1, False)]:
        for clamp, gain in [(None, None), (0.5, 2.0)]:
            x0 = torch.randn(shape)
            b0 = torch.randn(shape[dim])
            xq, bq = x0.to(dtype).float(), b0.to(dtype).float()      # same quantised inputs on both sides
            xr = xq.clone().requires_grad_(True); br = bq.clone().requires_grad_(True)
            yr = O.bias_act(xr, br, dim=dim, act=act, gain=gain, clamp=clamp)
            xg = xq.to(dev, dtype)
            if cl:
                xg = xg.contiguous(memory_format=torch.channels_last)
            xg.requires_grad_(True)
            bg = bq.to(dev, dtype).requires_grad_(True)
            yg = bias_act.bias_act(xg, bg, dim=dim, act=act, gain=gain, clamp=clamp)
            check(yg, yr, TOL[dtype], f"{act} y")
            dy = torch.randn(shape).to(dtype).float()
            gxr, gbr = torch.autograd.grad((yr * dy).sum(), [xr, br], create_graph=True)
            gxg, gbg = torch.autograd.grad((yg * dy.to(dev, dtype)).sum(), [xg, bg], create_graph=True)
            bshape = [1] * len(shape); bshape[dim] = -1
            nonzero = ((xq + bq.reshape(bshape)) != 0).float()   # derivative at exactly 0 is a convention (selu/relu kink)
            if dtype != torch.float32 and clamp is not None:
                # 16-bit storage rounds y onto the clamp boundary for a few elements, which flips their gradient mask
                # (the reference's fp16 path behaves the same): compare away from the boundary only.
                keep = ((yr.detach().abs() - clamp).abs() > 0.02 * clamp).float() * nonzero
                check(gxg * keep.to(dev), gxr * keep, TOL[dtype] * 2, f"{act} dx")
            elif float(nonzero.min()) == 0:
                check(gxg * nonzero.to(dev), gxr * nonzero, TOL[dtype] * 2, f"{act} dx")
            else:
                check(gxg, gxr, TOL[dtype] * 2, f"{act} dx")
                check(gbg, gbr, TOL[dtype] * 4, f"{act} db")
            if dtype == torch.float32 and gxr.requires_grad:      # second order (R1-style): d/dx of |dx|^2
                ggr = torch.autograd.grad(gxr.square().sum(), [xr], allow_unused=True)[0]
                ggg = torch.autograd.grad(gxg.square().sum(), [xg], allow_unused=True)[0] if gxg.requires_grad else None
                if ggr is None or float(ggr.abs().max()) == 0:
                    assert ggg is None or float(ggg.abs().max()) < 1e-6
                else:
                    check(ggg, ggr, 1e-4, f"{act} d2x")


def test_bias_act_no_bias_and_unaligned(dev):
    torch.manual_seed(2)
    x = torch.randn(3, 5, 7)
    y = bias_act.bias_act(x.to(dev), act="lrelu")
    check(y, O.bias_act(x, act="lrelu"), 1e-6)
    base = torch.randn(1001, device=dev)
    xs = base[1:]                       # 4-byte aligned only
    y = bias_act.bias_act(xs, act="swish", clamp=0.7)
    check(y, O.bias_act(xs.cpu(), act="swish", clamp=0.7), 1e-5)
    with pytest.raises(RuntimeError):
        bias_act.bias_act(torch.randn(4, 4), act="relu")      # CPU tensors are refused: no CPU path in the product


# ---------------------------------------------------------------------------------------------------- upfirdn2d

FILTERS = {
    "k4": [1, 3, 3, 1],
    "sym6": [0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633, 0.4910559419267466,
             0.787641141030194, 0.3379294217276218, -0.07263752278646252, -0.021060292512300564, 0.04472490177066578,
             0.0017677118642428036, -0.007800708325034148],
    "none": None,
    "k3x2": [[1., 2., 1.], [0.5, -1., 3.]],
}


@pytest.mark.parametrize("fname", list(FILTERS.keys()))
@pytest.mark.parametrize("updown", [(1, 1), (2, 1), (1, 2), (2, 2), (3, 2)])
def test_upfirdn2d(dev, fname, updown):
    torch.manual_seed(3)
    up, down = updown
    f = None if FILTERS[fname] is None else O.setup_filter(FILTERS[fname], normalize=(fname != "k3x2"))
    for shape, cl, dtype in [((2, 3, 8, 8), False, torch.float32), ((1, 4, 9, 9), False, torch.float32),
                             ((2, 16, 9, 7), True, torch.float32), ((2, 8, 10, 12), True, torch.bfloat16)]:
        for padding, flip, gain in [(0, False, 1), ([2, 1, 2, 1], True, 4), ([-1, 2, 3, -1], False, 1)]:
            fw, fh = O._fsize(f)
            px0, px1, py0, py1 = O._pad4(padding)
            if (shape[3] * up + px0 + px1 - fw) < 0 or (shape[2] * up + py0 + py1 - fh) < 0:
                continue
            xq = torch.randn(shape).to(dtype).float()
            xr = xq.clone().requires_grad_(True)
            yr = O.upfirdn2d(xr, f, up=up, down=down, padding=padding, flip_filter=flip, gain=gain)
            xg = xq.to(dev, dtype)
            if cl:
                xg = xg.contiguous(memory_format=torch.channels_last)
            xg.requires_grad_(True)
            fg = f.to(dev) if f is not None else None
            yg = upfirdn2d.upfirdn2d(xg, fg, up=up, down=down, padding=padding, flip_filter=flip, gain=gain)
            check(yg, yr, TOL[dtype], f"upfirdn {fname} {updown} {padding}")
            dy = torch.randn(yr.shape).to(dtype).float()
            dyr = dy.clone().requires_grad_(True)
            dyg = dy.to(dev, dtype).requires_grad_(True)
            gr = torch.autograd.grad((yr * dyr).sum(), xr, create_graph=True)[0]
            gg = torch.autograd.grad((yg * dyg).sum(), xg, create_graph=True)[0]
            check(gg, gr, TOL[dtype] * 2, f"upfirdn grad {fname} {updown} {padding}")
            # second order: the gradient op is itself differentiable w.r.t. dy (it is the forward op again)
            v = torch.randn(xq.shape).to(dtype).float()
            g2r = torch.autograd.grad((gr * v).sum(), dyr)[0]
            g2g = torch.autograd.grad((gg * v.to(dev, dtype)).sum(), dyg)[0]
            check(g2g, g2r, TOL[dtype] * 2, f"upfirdn grad-of-grad {fname} {updown} {padding}")


def test_upfirdn2d_matrix_core_fir(dev):
    """4x4 low-pass, up = down = 1, bf16 / f16 channel-minor tensors with C % 64 == 0: the matrix-core FIR kernel (Toeplitz MFMA),
    incl. every padding the layers use, crops, ragged tile edges, flip, gain and the fp32 fallback for inexact taps."""
    torch.manual_seed(12)
    f = O.setup_filter([1, 3, 3, 1])
    f_asym = torch.tensor([[1., 2., 4., 8.], [0.5, 1., 2., 4.], [3., 1., 0.25, 2.], [1., 1., 1., 1.]]) / 8     # exact in bf16, not symmetric
    f_inexact = f * 1.0001
    for dtype in [torch.bfloat16, torch.float16]:
        for (n, c, h, w, pad, gain, flip, ff) in [(2, 64, 33, 65, [1, 1, 1, 1], 4.0, False, f), (1, 128, 16, 32, [2, 2, 2, 2], 1.0, False, f),
                                                  (2, 64, 40, 24, [2, 1, 2, 1], 1.0, True, f_asym), (1, 192, 9, 70, [1, 2, 0, 3], 2.0, False, f_asym),
                                                  (1, 64, 20, 40, [-1, 2, 2, -1], 1.0, False, f), (1, 64, 17, 33, [1, 1, 1, 1], 1.0, False, f_inexact)]:
            x = torch.randn(n, c, h, w).to(dtype).float()
            ref = O.upfirdn2d(x, ff, padding=pad, gain=gain, flip_filter=flip)
            xg = x.to(dev, dtype).contiguous(memory_format=torch.channels_last)
            got = upfirdn2d.upfirdn2d(xg, ff.to(dev), padding=pad, gain=gain, flip_filter=flip)
            check(got, ref, TOL[dtype], f"mfma fir {dtype} c{c} {h}x{w} pad{pad}")


def test_upfirdn2d_wrappers(dev):
    torch.manual_seed(4)
    f = O.setup_filter([1, 3, 3, 1])
    x = torch.randn(2, 3, 16, 16)
    for name in ["filter2d", "upsample2d", "downsample2d"]:
        yr = getattr(O, name)(x, f)
        yg = getattr(upfirdn2d, name)(x.to(dev), f.to(dev))
        check(yg, yr, 1e-5, name)
    assert torch.equal(upfirdn2d.setup_filter([1, 3, 3, 1]), f)
    assert upfirdn2d.setup_filter(FILTERS["sym6"]).ndim == 1


# ---------------------------------------------------------------------------------------------------- conv

def _conv_case(dev, dtype, n, cin, cout, h, w, k, stride, pad, transpose, second_order=False):
    torch.manual_seed(5)
    xq = torch.randn(n, cin, h, w).to(dtype).float()
    wshape = (cin, cout, k, k) if transpose else (cout, cin, k, k)
    wq = (torch.randn(wshape) / np.sqrt(cin * k * k)).to(dtype).float()
    xr = xq.clone().requires_grad_(True); wr = wq.clone().requires_grad_(True)
    if transpose:
        yr = torch.nn.functional.conv_transpose2d(xr, wr, stride=stride, padding=pad)
    else:
        yr = torch.nn.functional.conv2d(xr, wr, stride=stride, padding=pad)
    xg = xq.to(dev, dtype).requires_grad_(True); wg = wq.to(dev, dtype).requires_grad_(True)
    fn = conv2d_gradfix.conv_transpose2d if transpose else conv2d_gradfix.conv2d
    yg = fn(xg, wg, stride=stride, padding=pad)
    tol = CONV_TOL[dtype]
    tag = f"conv n{n} {cin}->{cout} {h}x{w} k{k} s{stride} p{pad} T{int(transpose)} {dtype}"
    check(yg, yr, tol, tag + " y")
    dy = torch.randn(yr.shape).to(dtype).float()
    gxr, gwr = torch.autograd.grad((yr * dy).sum(), [xr, wr], create_graph=second_order)
    gxg, gwg = torch.autograd.grad((yg * dy.to(dev, dtype)).sum(), [xg, wg], create_graph=second_order)
    check(gxg, gxr, tol, tag + " dx")
    check(gwg, gwr, tol * 2, tag + " dw")
    if second_order:    # R1-style: gradient of |dx|^2 w.r.t. weights and of |dw|^2 w.r.t. x
        r2 = torch.autograd.grad(gxr.square().sum() + gwr.square().sum(), [xr, wr])
        g2 = torch.autograd.grad(gxg.square().sum() + gwg.square().sum(), [xg, wg])
        check(g2[0], r2[0], tol * 4, tag + " d2x")
        check(g2[1], r2[1], tol * 4, tag + " d2w")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_conv2d_shapes(dev, dtype):
    cases = [
        # n, cin, cout, h, w, k, stride, pad, transpose
        (2, 16, 32, 8, 8, 3, 1, 1, False),
        (1, 8, 8, 5, 7, 3, 1, 1, False),
        (3, 3, 24, 9, 9, 1, 1, 0, False),        # fromrgb-like: Cin = 3 (padded to 8 on the host)
        (2, 40, 3, 8, 8, 1, 1, 0, False),        # torgb-like: Cout = 3
        (2, 24, 16, 9, 9, 3, 2, 0, False),       # stride-2 (D down path)
        (2, 16, 24, 4, 4, 3, 2, 0, True),        # transposed stride-2 (G up path), out 9x9
        (2, 136, 200, 6, 6, 3, 1, 1, False),     # ragged channels (Cin % 32 != 0, Cout % 128 != 0)
        (1, 520, 72, 4, 4, 3, 1, 1, False),
        (2, 16, 16, 6, 6, 3, 1, 0, True),        # transposed stride-1
        (2, 16, 16, 7, 5, 3, 2, 1, False),
        (2, 16, 8, 5, 5, 4, 2, 1, True),         # 4x4 transposed (DCGAN-style)
    ]
    for c in cases:
        _conv_case(dev, dtype, *c)


def test_conv2d_double_backward(dev):
    _conv_case(dev, torch.float32, 2, 16, 24, 6, 6, 3, 1, 1, False, second_order=True)
    _conv_case(dev, torch.float32, 2, 16, 8, 9, 9, 3, 2, 0, False, second_order=True)
    _conv_case(dev, torch.float32, 2, 8, 16, 4, 4, 3, 2, 0, True, second_order=True)


def test_conv2d_split_k(dev):
    """few output tiles, long reduction: the gather kernel splits K over workgroups and a second kernel sums the fp32 slabs"""
    _conv_case(dev, torch.float32, 8, 512, 520, 4, 4, 3, 1, 1, False)        # the 4x4 block: 6 folded passes -> K = 9 x 3072
    _conv_case(dev, torch.float32, 2, 264, 136, 8, 8, 3, 1, 1, False)
    _conv_case(dev, torch.float32, 4, 128, 128, 4, 4, 3, 2, 0, True)         # transposed phases write strided outputs: no split
    # 16-bit outputs and fused epilogues split too (the slab reduction applies the epilogue and the cast)
    _conv_case(dev, torch.bfloat16, 8, 512, 512, 8, 8, 3, 1, 1, False)
    _conv_case(dev, torch.bfloat16, 8, 512, 256, 8, 8, 3, 2, 1, False)
    from style_big_gan_amd.torch_utils.ops import conv_bias_act
    torch.manual_seed(21)
    xq = torch.randn(4, 512, 8, 8).to(torch.bfloat16).float(); wq = (torch.randn(136, 512, 3, 3) / 70).to(torch.bfloat16).float(); bq = torch.randn(136).to(torch.bfloat16).float()
    for act, clamp in (("lrelu", 1.0), ("linear", None), ("relu", 0.3)):
        ref = O.bias_act(torch.nn.functional.conv2d(xq, wq, padding=1), bq, act=act, clamp=clamp)
        got = conv_bias_act.conv2d_bias_act(xq.to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last), wq.to(dev, torch.bfloat16),
                                            bq.to(dev, torch.bfloat16), padding=1, act=act, clamp=clamp)
        check(got, ref, 2e-2, f"split-K conv + fused {act}")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_conv_transpose_stride2_one_pass_kernel(dev, dtype):
    """stride-2 3x3 transposed convolutions whose phase grids are multiples of 8 x 32 (+ 1): conv_up2_kernel writes the 8 x 32 tiles of all
    four phases, the last row / column go through the gather launch; ragged channel counts on both sides; forward, dx (a stride-2
    convolution) and dw against the oracle.  The launch log must show the kernel was the one that ran."""
    from style_big_gan_amd import _lib
    _lib.prof_enable(True); _lib.prof_fetch()
    _conv_case(dev, dtype, 2, 64, 64, 8, 32, 3, 2, 0, True)          # one tile per image, 17 x 65 out
    _conv_case(dev, dtype, 1, 96, 72, 24, 32, 3, 2, 0, True)         # Cin = 1.5 slices, Cout = 64 + 8
    _conv_case(dev, dtype, 2, 128, 136, 16, 64, 3, 2, 0, True)       # two tiles across, three channel tiles
    _lib.prof_enable(False)
    codes = [r["dims"][6] for r in _lib.prof_fetch() if r["kind"] == "conv_igemm"]
    assert sum(c // 1000000 == 9 for c in codes) >= 3, codes


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_thin_channel_kernels_multi_tile(dev, dtype):
    """conv_thin_kernel / conv_wgrad_thin_kernel (csrc/conv_thin.hip, conv_wgrad.hip: <= 64 / <= 32 channels, ffhq_sg2.yaml's 512^2 / 1024^2 blocks)
    at sizes with MANY 16 x 16 tiles per image, ragged in both directions, several images: forward, dx and dw against the oracle for a
    3x3 / stride 1, a stride-2, a transposed stride-2 (four phase tables) and a 1x1 layer with 3 outputs; the launch log must show that the
    thin kernels were the ones that ran (code 6xxxxxx = conv_thin_kernel, 3xxxxxx = conv_wgrad_thin_kernel)."""
    from style_big_gan_amd import _lib
    _lib.prof_enable(True); _lib.prof_fetch()
    _conv_case(dev, dtype, 2, 16, 16, 96, 160, 3, 1, 1, False)        # 6 x 10 tiles per image
    _conv_case(dev, dtype, 2, 32, 16, 129, 257, 3, 2, 0, False)       # stride 2 -> 64 x 128; its dx is a transposed thin conv (16 -> 32)
    _conv_case(dev, dtype, 2, 32, 16, 48, 80, 3, 2, 0, True)          # transposed -> 97 x 161: four phases with different grids
    _conv_case(dev, dtype, 3, 16, 3, 96, 160, 1, 1, 0, False)         # ToRGB-like 1x1, Cout = 3
    _conv_case(dev, dtype, 1, 24, 40, 70, 45, 3, 1, 1, False)         # ragged channels (Cin 24 -> 1.5 fragments, Cout 40), odd image
    _conv_case(dev, dtype, 2, 16, 32, 33, 513, 3, 1, 1, False)        # one very wide strip: 33 column tiles, 3 row tiles (the last one row)
    _lib.prof_enable(False)
    rec = _lib.prof_fetch()
    thin_conv = [r for r in rec if r["kind"] == "conv_igemm" and r["dims"][6] // 1000000 == 6]
    thin_wgrad = [r for r in rec if r["kind"] == "conv_wgrad" and r["dims"][6] // 1000000 == 3]
    assert len(thin_conv) >= 10, [r["dims"] for r in rec if r["kind"] == "conv_igemm"]          # >= 5 forwards + their data gradients
    assert len(thin_wgrad) >= 5, [r["dims"] for r in rec if r["kind"] == "conv_wgrad"]


def test_conv_halo8_experimental_kernel(dev):
    """csrc/conv_halo8.hip (8 waves, a finished tile drains behind the next one; opt-in through the experiment word, bit 256 -- DESIGN.md section 4
    says why it is not the default): same launches, same results as the 12-wave halo kernel -- plain and with the fused tail, 8 x 32 and 16 x 16
    tiles, one and several tiles per workgroup, one and several channel tiles; the launch log must show code 5xxxxxx."""
    from style_big_gan_amd import _lib
    lib = _lib.load()
    torch.manual_seed(12)
    cases = [(8, 64, 128, 64, 128), (36, 64, 128, 48, 48), (3, 128, 256, 160, 160), (2, 192, 128, 128, 256)]      # n, cin, cout, h, w
    for n, cin, cout, h, w in cases:
        xq = torch.randn(n, cin, h, w).to(torch.bfloat16).float(); wq = (torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5)).to(torch.bfloat16).float()
        osc = torch.rand(n, cout) + 0.5; noise = torch.randn(n, 1, h, w); bias = torch.randn(cout)
        conv = torch.nn.functional.conv2d(xq, wq, padding=1)
        ref_tail = O.bias_act(conv * osc[:, :, None, None] + noise, bias, act="lrelu", gain=1.2, clamp=1.5)
        epi = conv2d_gradfix.Epilogue(oscale=osc.to(dev), noise=noise.to(dev), bias=bias.to(dev), act="lrelu", alpha=0.2, gain=1.2, clamp=1.5)
        xg = xq.to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last); wg = wq.to(dev, torch.bfloat16)
        base_plain = conv2d_gradfix._conv_forward(xg, wg, (1, 1), (1, 1))
        base_tail = conv2d_gradfix._conv_forward(xg, wg, (1, 1), (1, 1), epi=epi)
        lib.sbg_experiment_set(256)
        try:
            _lib.prof_enable(True); _lib.prof_fetch()
            got_plain = conv2d_gradfix._conv_forward(xg, wg, (1, 1), (1, 1))
            got_tail = conv2d_gradfix._conv_forward(xg, wg, (1, 1), (1, 1), epi=epi)
            _lib.prof_enable(False)
            codes = [r["dims"][6] // 1000000 for r in _lib.prof_fetch() if r["kind"] == "conv_igemm"]
        finally:
            lib.sbg_experiment_set(0)
        assert codes == [5, 5], (codes, n, cin, cout, h, w)
        check(got_plain, conv, 2e-2, "halo8 plain vs oracle"); check(got_tail, ref_tail, 2e-2, "halo8 tail vs oracle")
        assert torch.equal(got_plain, base_plain) and torch.equal(got_tail, base_tail), "halo8 differs from the 12-wave kernel"


def test_split_bf16_cat_dense_of_strided_view(dev):
    """sbg_split_bf16_cat_nd on a permuted weight view == the memory-order split followed by .contiguous(), bit for bit; the parts sum back
    to the fp32 value within 2^-24 relative"""
    torch.manual_seed(3)
    w = torch.randn(24, 13, 3, 3, device=dev)
    view = w.permute(2, 3, 0, 1).reshape(9, 24, 13)                        # [tap, cout, cin]: a strided view of the parameter
    assert not view.is_contiguous()
    for order in (conv2d_gradfix._ORDER6[1], (0, 1, 2)):
        for dim in (2, 1):
            dense = conv2d_gradfix._split_cat(view, dim, order, dense=True)
            plain = conv2d_gradfix._split_cat(view, dim, order).contiguous()
            assert dense.is_contiguous() and dense.shape == plain.shape and torch.equal(dense, plain)
    parts = conv2d_gradfix._split_cat(view, 2, (0, 1, 2), dense=True).float().reshape(9, 24, 3, 13).sum(2)
    assert float((parts - view).abs().max()) <= 2.0 ** -22 * float(view.abs().max())


def test_conv2d_large_k_and_many_pixels(dev):
    # enough pixels that the weight-gradient kernel splits the pixel axis across workgroups
    _conv_case(dev, torch.bfloat16, 4, 64, 64, 64, 64, 3, 1, 1, False)
    _conv_case(dev, torch.float32, 2, 32, 32, 40, 40, 3, 1, 1, False)
    _conv_case(dev, torch.bfloat16, 2, 128, 128, 32, 32, 1, 1, 0, False)
    # row-chunk LDS-DMA weight-gradient kernel: stride 1 and stride 2 (forward strided conv and transposed conv), ragged channels
    _conv_case(dev, torch.bfloat16, 2, 72, 40, 32, 64, 3, 1, 1, False)
    _conv_case(dev, torch.bfloat16, 2, 40, 72, 65, 65, 3, 2, 0, False)     # -> 32 x 32
    _conv_case(dev, torch.bfloat16, 2, 72, 40, 32, 32, 3, 2, 0, True)      # -> 65 x 65
    _conv_case(dev, torch.float32, 1, 16, 24, 65, 129, 3, 2, 0, False)     # -> 32 x 64


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv2d_resample(dev, dtype):
    torch.manual_seed(6)
    f = O.setup_filter([1, 3, 3, 1])
    for k, (up, down), flip_weight, groups in itertools.product([1, 3], [(1, 1), (2, 1), (1, 2)], [True, False], [1, 2]):
        xq = torch.randn(2, 16, 8, 8).to(dtype).float()
        wq = (torch.randn(24, 16 // groups, k, k) / np.sqrt(16 * k * k / groups)).to(dtype).float()
        xr = xq.clone().requires_grad_(True); wr = wq.clone().requires_grad_(True)
        yr = O.conv2d_resample(xr, wr, f=f, up=up, down=down, padding=k // 2, groups=groups, flip_weight=flip_weight)
        xg = xq.to(dev, dtype).requires_grad_(True); wg = wq.to(dev, dtype).requires_grad_(True)
        yg = conv2d_resample.conv2d_resample(xg, wg, f=f.to(dev), up=up, down=down, padding=k // 2, groups=groups, flip_weight=flip_weight)
        tag = f"resample k{k} up{up} down{down} flip{flip_weight} g{groups} {dtype}"
        check(yg, yr, CONV_TOL[dtype] * 2, tag)
        dy = torch.randn(yr.shape).to(dtype).float()
        gr = torch.autograd.grad((yr * dy).sum(), [xr, wr])
        gg = torch.autograd.grad((yg * dy.to(dev, dtype)).sum(), [xg, wg])
        check(gg[0], gr[0], CONV_TOL[dtype] * 2, tag + " dx")
        check(gg[1], gr[1], CONV_TOL[dtype] * 4, tag + " dw")


def test_fma(dev):
    torch.manual_seed(7)
    a = torch.randn(2, 4, 5, 5, requires_grad=True); b = torch.randn(2, 4, 1, 1, requires_grad=True); c = torch.randn(2, 1, 5, 5, requires_grad=True)
    ag, bg, cg = [t.detach().to(dev).requires_grad_(True) for t in (a, b, c)]
    yr = O.fma(a, b, c); yg = fma.fma(ag, bg, cg)
    check(yg, yr, 1e-6)
    gr = torch.autograd.grad(yr.square().sum(), [a, b, c]); gg = torch.autograd.grad(yg.square().sum(), [ag, bg, cg])
    for u, v in zip(gg, gr):
        check(u, v, 1e-5)


def test_scale_nc_gradients_in_one_pass(dev):
    """y = x * a[n, c] (the style modulation in front of every synthesis convolution, reference generators.py:79): first-order backward takes both
    gradients from ONE pass over (dy, x) (sbg_dot_hw_scale); they must be autograd's `dy * a` and `(dy * x).sum([2, 3])` -- checked against a
    float64 statement for 16-bit channel-minor tensors (the training layout), fp32, a channel count whose vectors do not tile a workgroup
    (falls back to the two kernels) and an image larger than one pixel split."""
    from style_big_gan_amd.torch_utils.ops import modulate
    torch.manual_seed(11)
    for dtype, n, c, h, w in ((torch.bfloat16, 3, 64, 20, 24), (torch.bfloat16, 2, 128, 96, 96), (torch.float32, 2, 32, 9, 7), (torch.bfloat16, 2, 24, 8, 8)):
        x = torch.randn(n, c, h, w).to(dtype); a = torch.randn(n, c) + 1; dy = torch.randn(n, c, h, w).to(dtype)
        xg = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True); ag = a.to(dev).requires_grad_(True)
        y = modulate.scale_nc(xg, ag)
        gx, ga = torch.autograd.grad(y, [xg, ag], dy.to(dev).contiguous(memory_format=torch.channels_last))
        x64, dy64, a64 = x.double(), dy.double(), a.double()
        tol = 1e-5 if dtype == torch.float32 else 1e-2
        check(y, (x64 * a64[:, :, None, None]).float(), tol, f"scale_nc {dtype} {c}")
        check(gx, (dy64 * a64[:, :, None, None]).float(), tol, f"scale_nc dx {dtype} {c}")
        check(ga, (dy64 * x64).sum([2, 3]).float(), 1e-5 if dtype == torch.float32 else 2e-3, f"scale_nc da {dtype} {c}")
    # the second-order path (composition of differentiable ops) still agrees with the fused first-order one
    xg = torch.randn(2, 64, 12, 12, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ag = (torch.randn(2, 64, device=dev) + 1).requires_grad_(True)
    dyg = torch.randn(2, 64, 12, 12, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    first = torch.autograd.grad(modulate.scale_nc(xg, ag), [xg, ag], dyg)
    second = torch.autograd.grad(modulate.scale_nc(xg, ag), [xg, ag], dyg, create_graph=True)
    check(first[0], second[0].detach().float().cpu(), 1e-6); check(first[1], second[1].detach().float().cpu(), 1e-4)


def test_conv_k64_halo_and_gather_kernels(dev):
    """shapes large enough (>= 256 tiles of 128 x 256) to take the persistent halo-staged kernels of csrc/conv_k64.hip
    -- (8, 32) patches with W % 32 == 0, (16, 16) patches otherwise -- and the 128 x 256 gather kernel, with ragged channel counts
    (Cin % 64 != 0, Cout % 128 != 0) and more tiles than workgroups; fp32 goes through the split passes with an accumulating epilogue."""
    _conv_case(dev, torch.bfloat16, 4, 72, 136, 64, 128, 3, 1, 1, False)       # halo (8, 32), 2 channel tiles, K tail
    _conv_case(dev, torch.bfloat16, 36, 64, 128, 48, 48, 3, 1, 1, False)       # halo (16, 16)
    _conv_case(dev, torch.bfloat16, 3, 128, 128, 160, 160, 3, 1, 1, False)     # 300 tiles on 256 workgroups: ragged last round
    _conv_case(dev, torch.float32, 2, 16, 128, 128, 256, 3, 1, 1, False)       # fp32: six bf16 passes, accumulate
    _conv_case(dev, torch.bfloat16, 2, 40, 200, 129, 257, 3, 2, 0, False)      # gather 128 x 256, stride 2 -> 64 x 128
    _conv_case(dev, torch.bfloat16, 2, 72, 40, 128, 128, 3, 2, 0, True)        # transposed: four gather launches -> 257 x 257
    # fused epilogue on the halo kernel vs the fp32 oracle composition
    torch.manual_seed(8)
    n, cin, cout, r = 4, 64, 128, 128
    xq = torch.randn(n, cin, r, r).to(torch.bfloat16).float(); wq = (torch.randn(cout, cin, 3, 3) / 24).to(torch.bfloat16).float()
    osc = torch.rand(n, cout) + 0.5; noise = torch.randn(n, 1, r, r); bias = torch.randn(cout)
    ref = torch.nn.functional.conv2d(xq, wq, padding=1) * osc[:, :, None, None] + noise
    ref = O.bias_act(ref, bias, act="lrelu", gain=1.2, clamp=1.5)
    epi = conv2d_gradfix.Epilogue(oscale=osc.to(dev), noise=noise.to(dev), bias=bias.to(dev), act="lrelu", alpha=0.2, gain=1.2, clamp=1.5)
    got = conv2d_gradfix._conv_forward(xq.to(dev, torch.bfloat16), wq.to(dev, torch.bfloat16), (1, 1), (1, 1), epi=epi)
    check(got, ref, 2e-2, "halo kernel fused epilogue vs oracle")


def test_fused_conv_bias_act_epilogue(dev):
    """conv + bias_act in one kernel == the two-op composition (forward, first and second order), incl. strided convs.
    Small-integer operands make every pre-activation exactly representable in bf16, so both paths see the same activation /
    clamp masks and must agree to rounding (with random reals a handful of near-zero pre-activations flip sign between the two
    roundings, which moves individual gradient elements by ~10 % without either path being wrong)."""
    from style_big_gan_amd.torch_utils.ops import conv_bias_act
    torch.manual_seed(9)
    for (stride, pad, act, clamp, gain) in [(1, 1, "lrelu", 40.0, None), (2, 0, "lrelu", None, 0.5), (1, 0, "linear", 6.0, None), (1, 1, "relu", None, None)]:
        k = 1 if pad == 0 and stride == 1 else 3
        x = torch.randint(-2, 3, (2, 8, 17, 17), device=dev).to(torch.bfloat16).requires_grad_(True)
        w = torch.randint(-1, 2, (40, 8, k, k), device=dev).to(torch.bfloat16).requires_grad_(True)
        b = (torch.randint(-3, 4, (40,), device=dev).float() + 0.5).to(torch.bfloat16).requires_grad_(True)      # never lands on 0
        y_f = conv_bias_act.conv2d_bias_act(x, w, b, stride=stride, padding=pad, act=act, gain=gain, clamp=clamp)
        y_u = bias_act.bias_act(conv2d_gradfix.conv2d(x, w, stride=stride, padding=pad), b, act=act, gain=gain, clamp=clamp)
        assert rel_err(y_f, y_u) < 1e-2
        dy = torch.randn_like(y_f)
        gf = torch.autograd.grad((y_f * dy).sum(), [x, w, b], create_graph=True)
        gu = torch.autograd.grad((y_u * dy).sum(), [x, w, b], create_graph=True)
        for a_, b_ in zip(gf, gu):
            assert rel_err(a_, b_) < 2e-2
        g2f = torch.autograd.grad(gf[0].float().square().sum(), w)[0]     # R1-style second order
        g2u = torch.autograd.grad(gu[0].float().square().sum(), w)[0]
        assert rel_err(g2f, g2u) < 3e-2
    # random reals against the fp32 oracle (forward)
    xq = torch.randn(2, 16, 8, 8).to(torch.bfloat16).float(); wq = (torch.randn(24, 16, 3, 3) / 12).to(torch.bfloat16).float(); bq = torch.randn(24).to(torch.bfloat16).float()
    ref = O.bias_act(torch.nn.functional.conv2d(xq, wq, padding=1), bq, act="lrelu", clamp=1.0)
    got = conv_bias_act.conv2d_bias_act(xq.to(dev, torch.bfloat16), wq.to(dev, torch.bfloat16), bq.to(dev, torch.bfloat16), padding=1, act="lrelu", clamp=1.0)
    check(got, ref, 2e-2, "fused conv+bias_act vs oracle")


def test_fused_modconv_training_layer(dev):
    """SynthesisLayer body: fused epilogue + one-pass backward head (ops/modconv.py) vs the modulated_conv2d + bias_act composition.
    Both run in bf16 and individual activation / clamp masks flip between any two bf16 evaluation orders, so each path is held to
    an fp64 restatement of the reference layer (generators.py:79-88,328) on the same inputs and the same dy: the fused path must be
    at least as close as the composition (x1.5 + 5e-3 slack) for y and every first-order gradient."""
    from style_big_gan_amd.torch_utils.ops import modconv
    from style_big_gan_amd.train_parts import generators as GN
    F = torch.nn.functional
    torch.manual_seed(11)

    def ref_layer(x, w, s, nz, b, act, gain, clamp):
        dco = ((w[None] * s[:, None, :, None, None]).square().sum([2, 3, 4]) + 1e-8).rsqrt()
        pre = F.conv2d(x * s[:, :, None, None], w, padding=1) * dco[:, :, None, None] + (nz if nz is not None else 0) + b[None, :, None, None]
        y = (F.leaky_relu(pre, 0.2) if act == "lrelu" else pre) * gain
        return y.clamp(-clamp, clamp) if clamp is not None else y

    def err(a, b):
        return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-12))

    for (n, cin, cout, r, act, clamp, noise_kind) in [(2, 16, 128, 16, "lrelu", 2.0, "per_sample"), (3, 24, 64, 8, "lrelu", None, "const"),
                                                      (2, 8, 128, 16, "linear", 1.0, None), (4, 64, 128, 32, "lrelu", None, "per_sample")]:
        x0 = torch.randn(n, cin, r, r, device=dev).to(torch.bfloat16)
        w0 = torch.randn(cout, cin, 3, 3, device=dev).to(torch.bfloat16).float()
        s0 = torch.randn(n, cin, device=dev) + 1
        b0 = torch.randn(cout, device=dev)
        nz0 = None if noise_kind is None else torch.randn((n, 1, r, r) if noise_kind == "per_sample" else (r, r), device=dev)
        dy0 = torch.randn(n, cout, r, r, device=dev).to(torch.bfloat16)
        res = {}
        for mode in ("fused", "unfused", "ref"):
            cast = (lambda t: t.double()) if mode == "ref" else (lambda t: t.clone())
            x, w, s, b = [cast(t).requires_grad_(True) for t in (x0, w0, s0, b0)]
            nz = None if nz0 is None else cast(nz0).requires_grad_(True)
            if mode == "fused":
                assert modconv.usable(x, w, act, 1)
                y = modconv.modconv_bias_act(x, w.to(x.dtype), s, GN.demod_coefficients(w, s), nz, b, padding=1, act=act, gain=1.3, clamp=clamp)
            elif mode == "unfused":
                y = bias_act.bias_act(GN.modulated_conv2d(x=x, weight=w, styles=s, noise=nz, padding=1), b.to(torch.bfloat16), act=act, gain=1.3, clamp=clamp)
            else:
                y = ref_layer(x, w, s, nz, b, act, 1.3, clamp)
            g = torch.autograd.grad((y.double() * dy0.double()).sum(), [x, w, s, b] + ([nz] if nz is not None else []))
            res[mode] = [y.detach()] + [t.detach() for t in g]
        for name, f_, u_, r_ in zip(["y", "dx", "dw", "dstyles", "db", "dnoise"], res["fused"], res["unfused"], res["ref"]):
            ef, eu = err(f_, r_), err(u_, r_)
            assert ef <= 1.5 * eu + 5e-3, f"modconv {name} ({act}, {noise_kind}, clamp {clamp}): fused {ef:.3e} vs composition {eu:.3e}"


def test_fused_up_synthesis_layer(dev):
    """Up-sampling SynthesisLayer (x * s -> multi-phase transposed conv -> low-pass with the fused tail, ops/upfirdn2d.py::_FirBiasAct)
    vs the unfused bf16 composition, both held to an fp64 evaluation of the oracle's layer (oracle/ops.py::modulated_conv2d + bias_act):
    the fused path must be at least as close (x1.5 + 5e-3) for y and every first-order gradient."""
    from style_big_gan_amd.torch_utils.ops import modconv
    from style_big_gan_amd.train_parts import generators as GN
    torch.manual_seed(13)
    f = O.setup_filter([1, 3, 3, 1])

    def err(a, b):
        return float((a.double().cpu() - b.double().cpu()).abs().max() / (b.double().cpu().abs().max() + 1e-12))

    for (n, cin, cout, r, noise_kind, clamp) in [(2, 32, 64, 16, "per_sample", 2.0), (3, 16, 128, 8, "const", None)]:
        layer = GN.SynthesisLayer(cin, cout, w_dim=24, resolution=2 * r, up=2, use_noise=True, conv_clamp=clamp, channels_last=True).to(dev)
        with torch.no_grad():
            layer.noise_strength.fill_(0.3); layer.bias.normal_()
        x0 = torch.randn(n, cin, r, r, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w0 = torch.randn(n, 24, device=dev)
        dy0 = torch.randn(n, cout, 2 * r, 2 * r, device=dev).to(torch.bfloat16)
        noise_mode = "const" if noise_kind == "const" else "random"
        res = {}
        for mode in ("fused", "unfused"):
            modconv.enabled = (mode == "fused")
            try:
                x = x0.clone().requires_grad_(True); w = w0.clone().requires_grad_(True)
                torch.manual_seed(99)                              # same random noise in both runs
                y = layer(x, w, noise_mode=noise_mode)
                assert ("FirBiasAct" in type(y.grad_fn).__name__) == (mode == "fused"), type(y.grad_fn).__name__
                params = [layer.weight, layer.bias, layer.noise_strength, layer.affine.weight]
                g = torch.autograd.grad((y.double() * dy0.double()).sum(), [x, w] + params)
            finally:
                modconv.enabled = True
            res[mode] = [y.detach()] + [t.detach() for t in g]
        # fp64 reference of the same layer on the CPU oracle ops
        torch.manual_seed(99)
        noise = (torch.randn([n, 1, 2 * r, 2 * r], device=dev) if noise_mode == "random" else layer.noise_const.detach()).double().cpu()
        xr = x0.double().cpu().requires_grad_(True); wr = w0.double().cpu().requires_grad_(True)
        pw = [p.detach().double().cpu().requires_grad_(True) for p in (layer.weight, layer.bias, layer.noise_strength, layer.affine.weight)]
        styles = torch.addmm(layer.affine.bias.detach().double().cpu().unsqueeze(0), wr, (pw[3] * layer.affine.weight_gain).t())
        yr = O.modulated_conv2d(xr, pw[0], styles, noise=noise * pw[2], up=2, padding=1, resample_filter=f, flip_weight=False, fused_modconv=False)
        yr = O.bias_act(yr, pw[1], act="lrelu", gain=layer.act_gain, clamp=clamp)
        gr = torch.autograd.grad((yr * dy0.double().cpu()).sum(), [xr, wr] + pw)
        ref = [yr.detach()] + [t.detach() for t in gr]
        for name, f_, u_, r_ in zip(["y", "dx", "dw_latent", "dweight", "dbias", "dnoise_strength", "daffine"], res["fused"], res["unfused"], ref):
            ef, eu = err(f_, r_), err(u_, r_)
            assert ef <= 1.5 * eu + 5e-3, f"up layer {name} ({noise_kind}, clamp {clamp}): fused {ef:.3e} vs composition {eu:.3e}"


def test_master_weight_convolutions(dev):
    """fp32 parameter + 16-bit activations ("mixed" launches): same forward bits as casting `w * gain` first, gradients against fp64,
    R1-style double backward, the packed-operand cache following in-place parameter updates, channels_last parameters."""
    torch.manual_seed(11)
    for (cin, cout, k, stride, transpose, cl) in [(16, 24, 3, 1, False, False), (24, 16, 3, 2, False, True), (16, 24, 3, 2, True, False), (8, 3, 1, 1, False, False)]:
        gain = 0.37
        x = torch.randn(2, cin, 12, 12, device=dev)
        wshape = [cin, cout, k, k] if transpose else [cout, cin, k, k]
        w = torch.nn.Parameter(torch.randn(wshape, device=dev).to(memory_format=torch.channels_last if cl else torch.contiguous_format))
        xb = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        fn = conv2d_gradfix.conv_transpose2d if transpose else conv2d_gradfix.conv2d
        y_mixed = fn(xb, w, stride=stride, padding=k // 2, wgain=gain)
        y_cast = fn(xb, (w * gain).to(torch.bfloat16), stride=stride, padding=k // 2)
        assert torch.equal(y_mixed, y_cast)
        dy = torch.randn_like(y_mixed)
        dx, dw = torch.autograd.grad((y_mixed * dy).sum(), [xb, w])
        assert dw.dtype == torch.float32 and dw.shape == w.shape and dw.stride() == w.stride()
        x64 = xb.detach().double().requires_grad_(True)
        w64 = (w.detach() * gain).to(torch.bfloat16).double().requires_grad_(True)
        f64 = torch.nn.functional.conv_transpose2d if transpose else torch.nn.functional.conv2d
        y64 = f64(x64.cpu(), w64.cpu(), stride=stride, padding=k // 2)
        dx64, dw64 = torch.autograd.grad((y64 * dy.double().cpu()).sum(), [x64, w64])
        assert float((dw.double().cpu() - dw64.cpu() * gain).abs().max() / (dw64.abs().max() * gain)) < 2e-3       # fp32 accumulate of bf16 products
        assert float((dx.double().cpu() - dx64.cpu()).abs().max() / dx64.abs().max()) < 2e-2
        # cache follows the parameter's version counter
        with torch.no_grad():
            w.mul_(2.0)
        assert torch.equal(fn(xb, w, stride=stride, padding=k // 2, wgain=gain), fn(xb, (w * gain).to(torch.bfloat16), stride=stride, padding=k // 2))
    # fused optimizers update parameters without touching the version counter: the cache must still notice (per-parameter epoch)
    w = torch.nn.Parameter(torch.randn(16, 16, 3, 3, device=dev))
    xb = torch.randn(2, 16, 8, 8, device=dev).to(torch.bfloat16)
    for fused in (True, False):
        opt = torch.optim.Adam([w], lr=0.5, fused=fused)
        y0 = conv2d_gradfix.conv2d(xb, w, padding=1)
        w.grad = torch.randn_like(w)
        opt.step()
        y1 = conv2d_gradfix.conv2d(xb, w, padding=1)
        assert torch.equal(y1, conv2d_gradfix.conv2d(xb, w.detach().to(torch.bfloat16), padding=1)) and not torch.equal(y0, y1)
    # double backward through the mixed path (what R1 does with the discriminator's convolutions)
    w = torch.nn.Parameter(torch.randn(8, 8, 3, 3, device=dev))
    xb = torch.randn(2, 8, 8, 8, device=dev).to(torch.bfloat16).requires_grad_(True)
    for wt, kw in ((w, dict(wgain=0.5)), ((w * 0.5).to(torch.bfloat16), {})):
        y = conv2d_gradfix.conv2d(xb, wt, padding=1, **kw)
        g, = torch.autograd.grad(y.float().square().sum(), xb, create_graph=True)
        gw, = torch.autograd.grad(g.float().square().sum(), w)
        if kw:
            ref_gw = gw
    assert float((ref_gw - gw).abs().max() / gw.abs().max()) < 3e-2


def test_demod_coefficients_kernel(dev):
    from style_big_gan_amd.torch_utils.ops import modconv
    from style_big_gan_amd.train_parts.generators import demod_coefficients
    torch.manual_seed(12)
    for (n, o, i, k) in [(4, 32, 24, 3), (3, 512, 512, 3), (2, 16, 64, 1)]:
        w = torch.nn.Parameter(torch.randn(o, i, k, k, device=dev) * 0.1)
        s = (torch.randn(n, i, device=dev) + 1).requires_grad_(True)
        d_ref = demod_coefficients(w, s)                                   # torch composition (arbitrarily differentiable)
        d = modconv.demod_coefs(w, s, torch.bfloat16)
        assert float((d - d_ref).abs().max() / d_ref.abs().max()) < 1e-5
        g = torch.randn_like(d)
        gw_ref, gs_ref = torch.autograd.grad((d_ref * g).sum(), [w, s])
        gw, gs = torch.autograd.grad((d * g).sum(), [w, s])
        assert float((gw - gw_ref).abs().max() / gw_ref.abs().max()) < 1e-4 and float((gs - gs_ref).abs().max() / gs_ref.abs().max()) < 1e-4


def test_torgb_streaming_kernels(dev):
    """ops/torgb.py vs the fp64 composition clamp(conv1x1(x * s, w) + b): output, dx, d wmod (-> dw, ds), db; clamp mask; ragged pixel counts"""
    from style_big_gan_amd.torch_utils.ops import torgb
    torch.manual_seed(13)
    # (128 / 256 channels with a multiple of 16 pixels: the matrix-core forward, weights in three 16-bit parts; the others: the streaming kernel)
    for (n, c, o, h, w, clamp) in [(3, 128, 3, 16, 16, 0.8), (2, 256, 3, 24, 32, 1.5), (2, 128, 2, 12, 20, None), (2, 512, 3, 8, 8, None), (2, 64, 3, 17, 13, 0.5), (4, 32, 1, 9, 9, 256.0), (2, 8, 4, 5, 7, None)]:
        x = torch.randn(n, c, h, w, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wt = (torch.randn(o, c, 1, 1, device=dev) / c ** 0.5).requires_grad_(True)
        s = (torch.randn(n, c, device=dev) * 0.5 + 1).requires_grad_(True)
        b = torch.randn(o, device=dev).requires_grad_(True)
        assert torgb.usable(x, wt)
        wmod = wt.reshape(1, o, c) * s.unsqueeze(1)
        y = torgb.torgb(x, wmod, b, clamp=clamp)
        assert y.dtype == torch.float32 and y.shape == (n, o, h, w) and y.is_contiguous()
        dy = torch.randn_like(y)
        gx, gw, gs, gb = torch.autograd.grad((y * dy).sum(), [x, wt, s, b])
        x64, w64, s64, b64 = (t.detach().double().cpu().requires_grad_(True) for t in (x, wt, s, b))
        pre = torch.einsum('nchw,oc,nc->nohw', x64, w64[:, :, 0, 0], s64) + b64.reshape(1, -1, 1, 1)
        y64 = pre.clamp(-clamp, clamp) if clamp is not None else pre
        r = torch.autograd.grad((y64 * dy.double().cpu()).sum(), [x64, w64, s64, b64])
        rel = lambda a, ref: float((a.double().cpu() - ref).abs().max() / (ref.abs().max() + 1e-12))
        assert rel(y, y64) < 1e-5, (c, rel(y, y64))
        assert rel(gx, r[0]) < 1e-2 and rel(gw, r[1]) < 1e-4 and rel(gs, r[2]) < 1e-4 and rel(gb, r[3]) < 1e-5, (c, rel(gx, r[0]), rel(gw, r[1]), rel(gs, r[2]))


def test_fromrgb_streaming_kernels(dev):
    """ops/fromrgb.py vs the fp64 composition clamp(lrelu(conv1x1(img, w * gain) + b) * act_gain): output; dimg, dw, db against fp64 sums
    under the op's own gradient convention (masks read from the stored 16-bit output); ragged pixel counts"""
    from style_big_gan_amd.torch_utils.ops import fromrgb
    torch.manual_seed(14)
    old = fromrgb.enabled
    fromrgb.enabled = True
    try:
        for (n, ci, co, h, w, act, clamp) in [(3, 3, 128, 16, 16, "lrelu", 256.0), (2, 3, 512, 8, 8, "lrelu", 0.7), (2, 1, 64, 17, 13, "relu", None),
                                              (2, 3, 32, 9, 31, "linear", 0.5), (1, 4, 8, 5, 7, "lrelu", None)]:
            img = torch.randn(n, ci, h, w, device=dev).requires_grad_(True)
            wt = torch.randn(co, ci, 1, 1, device=dev).requires_grad_(True)
            b = torch.randn(co, device=dev).requires_grad_(True)
            wg = 1 / np.sqrt(ci)
            assert fromrgb.usable(img, wt, act, torch.bfloat16)
            y = fromrgb.fromrgb(img, wt, b, wg, act, clamp=clamp)
            assert y.dtype == torch.bfloat16 and y.shape == (n, co, h, w) and y.is_contiguous(memory_format=torch.channels_last)
            dy = torch.randn(n, co, h, w, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            gi, gw, gb = torch.autograd.grad((y.float() * dy.float()).sum(), [img, wt, b])
            i64, w64, b64 = (t.detach().double().cpu() for t in (img, wt, b))
            pre = torch.einsum('nchw,oc->nohw', i64, w64[:, :, 0, 0] * wg) + b64.reshape(1, -1, 1, 1)
            a, g = {"lrelu": (0.2, np.sqrt(2)), "relu": (0.0, np.sqrt(2)), "linear": (None, 1.0)}[act]
            z = pre if a is None else torch.where(pre > 0, pre, pre * a)
            y64 = z * g
            if clamp is not None:
                y64 = y64.clamp(-clamp, clamp)
            rel = lambda t, ref: float((t.double().cpu() - ref).abs().max() / (ref.abs().max() + 1e-12))
            assert rel(y.detach(), y64) < 1e-2                                     # bf16 output
            # gradient convention (bias_act.cu:141, applied to the stored 16-bit output): slope and rail test read the saved y
            ys = y.detach().double().cpu()
            slope = torch.full_like(ys, g) if a is None else torch.where(ys > 0, torch.full_like(ys, g), torch.full_like(ys, g * a))
            if clamp is not None:
                rail = float(torch.tensor(clamp).to(torch.bfloat16))
                slope = torch.where(ys.abs() < rail, slope, torch.zeros_like(slope))
            d1 = dy.double().cpu() * slope
            r_i = torch.einsum('nohw,oc->nchw', d1, w64[:, :, 0, 0] * wg)
            r_w = (torch.einsum('nohw,nchw->oc', d1, i64) * wg).reshape(co, ci, 1, 1)
            r_b = d1.sum([0, 2, 3])
            assert rel(gw, r_w) < 1e-4 and rel(gb, r_b) < 1e-4 and rel(gi, r_i) < 1e-4, (co, act, rel(gw, r_w), rel(gb, r_b), rel(gi, r_i))
    finally:
        fromrgb.enabled = old


def test_maximum_sizes_and_empty_inputs(dev):
    """The reference's plugins limit tensors to INT_MAX elements (bias_act.cpp:40, upfirdn2d.cpp:22-23,36).  Same limit here, byte offsets
    64-bit: a bias_act just under the limit (4.3 GB of bf16) is spot-checked slice by slice, one element more is refused; empty batches
    pass through every op."""
    torch.manual_seed(15)
    n_el = 2 ** 31 - 8                                  # just under INT_MAX, multiple of 8
    c, w = 8, 4096
    h = n_el // (c * w)
    x = torch.empty([1, c, h, w], dtype=torch.bfloat16, device=dev)
    for i in range(0, h, 8192):                         # fill in slabs (randn_like on the whole tensor would need 8.6 GB of fp32)
        x[:, :, i:i + 8192] = torch.randn([1, c, min(8192, h - i), w], device=dev).to(torch.bfloat16)
    b = torch.randn(c, device=dev).to(torch.bfloat16)
    assert x.numel() <= 2 ** 31 - 1
    y = bias_act.bias_act(x, b, act="lrelu", clamp=1.5)
    for sl in (slice(0, 3), slice(h // 2, h // 2 + 3), slice(h - 3, h)):       # first rows, the middle (past 2 GiB), the very end
        ref = O.bias_act(x[:, :, sl].float().cpu(), b.float().cpu(), act="lrelu", clamp=1.5)
        check(y[:, :, sl], ref, 2e-2)
    del y
    with pytest.raises(RuntimeError, match="too large"):
        bias_act.bias_act(torch.empty([2 ** 31 + 8], dtype=torch.bfloat16, device=dev), act="relu")
    del x
    torch.cuda.empty_cache()
    # empty batches
    f = upfirdn2d.setup_filter([1, 3, 3, 1], device=dev)
    assert upfirdn2d.upfirdn2d(torch.zeros([0, 8, 16, 16], device=dev), f, up=2, padding=1).shape[0] == 0
    assert bias_act.bias_act(torch.zeros([0, 8, 4, 4], device=dev), torch.zeros(8, device=dev), act="lrelu").shape == (0, 8, 4, 4)
    assert fma.fma(torch.zeros([0, 4, 2, 2], device=dev), torch.zeros([0, 4, 1, 1], device=dev), torch.zeros([0, 1, 2, 2], device=dev)).numel() == 0


def test_separable_fused_kernel_against_two_passes(dev):
    """sbg_upfirdn2d_separable (both passes of a rank-1 filter in one launch, planar fp32) against the oracle over tap counts 1..16,
    up / down in {1, 2}, positive / negative / asymmetric padding, both flip settings, ragged image sizes; plus its gradient."""
    torch.manual_seed(16)
    cases = 0
    for taps in (1, 2, 3, 5, 8, 12, 16):
        f = torch.randn(taps)
        for up, down in ((1, 1), (2, 1), (1, 2), (2, 2)):
            for pad in (0, [taps // 2, (taps - 1) // 2, taps // 2, (taps - 1) // 2], [3, 1, 0, 2], [-1, 2, 1, -2]):
                for flip in (False, True):
                    n, c, h, w = 2, 3, 37, 70
                    x = torch.randn(n, c, h, w)
                    try:
                        ref = O.upfirdn2d(x, f, up=up, down=down, padding=pad, flip_filter=flip, gain=1.7)
                    except Exception:
                        continue                              # configuration with an empty output
                    if ref.numel() == 0:
                        continue
                    xg = x.to(dev).requires_grad_(True)
                    y = upfirdn2d.upfirdn2d(xg, f.to(dev), up=up, down=down, padding=pad, flip_filter=flip, gain=1.7)
                    check(y, ref, 1e-5)
                    if taps in (3, 12) and flip is False:
                        dy = torch.randn_like(ref)
                        xr = x.clone().requires_grad_(True)
                        gr, = torch.autograd.grad((O.upfirdn2d(xr, f, up=up, down=down, padding=pad, flip_filter=flip, gain=1.7) * dy).sum(), xr)
                        gg, = torch.autograd.grad((y * dy.to(dev)).sum(), xg)
                        check(gg, gr, 1e-5)
                    cases += 1
    assert cases > 150


def test_minibatch_std_kernels(dev):
    """csrc/mbstd.hip (forward incl. the copy of x, first-order backward) against the layer as tensor ops on the CPU in float64 (the reference's
    composition, discriminators.py:316-328): 1e-5; second order (R1 through the layer) through the composite fallback"""
    from style_big_gan_amd.train_parts.discriminators import MinibatchStdLayer, _minibatch_std_composite
    torch.manual_seed(5)
    for (n, c, hw, group, f) in [(32, 512, 4, 32, 1), (8, 64, 4, 4, 1), (12, 48, 4, 4, 3), (6, 16, 8, None, 2), (4, 8, 4, 8, 1)]:
        x = torch.randn(n, c, hw, hw) * 2 + 0.3
        layer = MinibatchStdLayer(group, f)
        g_ = min(int(group), n) if group is not None else n
        xr = x.double().requires_grad_(True)
        yr = _minibatch_std_composite(xr, g_, f)
        w = torch.randn_like(yr)
        (gr,) = torch.autograd.grad((yr * w).sum(), xr)
        xd = x.to(dev).requires_grad_(True)
        yd = layer(xd)
        assert yd.shape == yr.shape
        (gd,) = torch.autograd.grad((yd * w.to(dev).float()).sum(), xd)
        assert rel_err(yd, yr.float()) < 1e-5 and rel_err(gd, gr.float()) < 1e-5, (n, c, hw, group, f, rel_err(yd, yr.float()), rel_err(gd, gr.float()))
    # second order
    x = torch.randn(8, 16, 4, 4)
    def second(t, fn):
        (g1,) = torch.autograd.grad(fn(t).square().sum(), t, create_graph=True)
        return torch.autograd.grad(g1.square().sum(), t)[0]
    layer = MinibatchStdLayer(4, 1)
    ref = second(x.double().requires_grad_(True), lambda t: _minibatch_std_composite(t, 4, 1))
    got = second(x.to(dev).requires_grad_(True), layer)
    assert rel_err(got, ref.float()) < 1e-4


def test_fused_conv_bias_act_second_order_vs_oracle(dev):
    """the R1 pattern through the fused convolution + bias + activation op (what the discriminator's layers run in bf16): gradient of
    |d sum(y) / dx|^2 with respect to the weight and the bias, against the oracle's composition (F.conv2d + oracle bias_act) in float64 on the
    CPU with the same bf16-representable inputs.  lrelu with gain, with and without clamp, stride 1 and 2.  Tolerance 3e-2 of the gradient's
    max magnitude (bf16 activations; the comparison is with the ORACLE, not with the unfused HIP path)."""
    from style_big_gan_amd.torch_utils.ops import conv_bias_act
    torch.manual_seed(11)
    for stride, pad, clamp in [(1, 1, None), (1, 1, 2.0), (2, 1, None)]:
        x = (torch.randn(2, 16, 12, 12)).to(torch.bfloat16)
        w = (torch.randn(24, 16, 3, 3) / 12).to(torch.bfloat16)
        b = (torch.randn(24) * 0.3).to(torch.bfloat16)

        def r1_like(xx, ww, bb, fwd):
            y = fwd(xx, ww, bb)
            (gx,) = torch.autograd.grad(y.float().sum() if y.dtype != torch.float64 else y.sum(), xx, create_graph=True)
            pen = gx.float().square().sum() if gx.dtype != torch.float64 else gx.square().sum()
            return y, torch.autograd.grad(pen, [ww, bb], allow_unused=True)

        xr, wr, br = [t.double().requires_grad_(True) for t in (x, w, b)]
        yr, gr = r1_like(xr, wr, br, lambda a_, w_, b_: O.bias_act(torch.nn.functional.conv2d(a_, w_, stride=stride, padding=pad), b_, act="lrelu", gain=1.3, clamp=clamp))
        xd, wd, bd = [t.to(dev).requires_grad_(True) for t in (x, w, b)]
        yd, gd = r1_like(xd, wd, bd, lambda a_, w_, b_: conv_bias_act.conv2d_bias_act(a_, w_, b_, stride=stride, padding=pad, act="lrelu", gain=1.3, clamp=clamp))
        check(yd, yr.float(), 2e-2, f"fused y (stride {stride}, clamp {clamp})")
        check(gd[0], gr[0].float(), 3e-2, f"fused d2w (stride {stride}, clamp {clamp})")
        if gr[1] is not None and float(gr[1].abs().max()) > 0:
            check(gd[1] if gd[1] is not None else torch.zeros_like(bd), gr[1].float(), 3e-2, f"fused d2b (stride {stride}, clamp {clamp})")


def test_grouped_gemm_against_torch(dev):
    """sbg_grouped_gemm (csrc/grouped_gemm.hip): a table of small fp32 products in one launch -- transposed and sliced operands by strides, one and two
    terms, bias, row sums, ragged sizes, more problems than one launch's table holds (16) -- against torch.matmul in fp64."""
    from style_big_gan_amd.torch_utils.ops import grouped_gemm
    torch.manual_seed(0)
    probs, refs = [], []
    big = torch.randn(70, 9, 96, device=dev)
    for i in range(37):
        m, n, k = [(64, 128, 512), (70, 33, 96), (1, 512, 17), (130, 5, 64), (16, 64, 1)][i % 5]
        a = torch.randn(m, k, device=dev) if i % 3 else torch.randn(k, m, device=dev).t()
        b = torch.randn(k, n, device=dev) if i % 2 else torch.randn(n, k, device=dev).t()
        if (m, k) == (70, 96):
            a = big[:, i % 9]                                        # a slice of a [N, L, D] tensor (row stride L * D)
        terms, ref = [(a, b, 0.37)], 0.37 * (a.double() @ b.double())
        if i % 4 == 1:
            k2 = 40
            a2, b2 = torch.randn(m, k2, device=dev), torch.randn(k2, n, device=dev)
            terms.append((a2, b2, -1.5)); ref = ref - 1.5 * (a2.double() @ b2.double())
        c = torch.full([m, n + 3], float('nan'), device=dev)[:, 1:n + 1] if i % 5 == 2 else torch.full([m, n], float('nan'), device=dev)
        pr = dict(c=c, terms=terms)
        if i % 2 == 0:
            pr['bias'] = torch.randn(n, device=dev); pr['bias_scale'] = 0.5
            ref = ref + 0.5 * pr['bias'].double()
        rs_ref = None
        if i % 3 == 0:
            pr['rowsum'] = torch.full([m], float('nan'), device=dev); pr['rowsum_scale'] = 2.0
            rs_ref = 2.0 * a.double().sum(1)
        probs.append(pr); refs.append((ref, rs_ref))
    grouped_gemm.launch(probs, dev)
    for pr, (ref, rs_ref) in zip(probs, refs):
        tol = 1e-5 * max(1.0, float(ref.abs().max()))
        assert float((pr['c'].double() - ref).abs().max()) < tol
        if rs_ref is not None:
            assert float((pr['rowsum'].double() - rs_ref).abs().max()) < 1e-5 * max(1.0, float(rs_ref.abs().max()))
    grouped_gemm.launch([], dev)


@pytest.mark.parametrize("nfp", [0, 2])
def test_style_bank_equals_per_layer_affines(dev, nfp):
    """SynthesisNetwork._style_bank (every layer's styles of a pass from one launch, backward from two) against the per-layer FullyConnectedLayer calls it
    replaces (reference generators.py:333, 397): image, gradient of ws and of every affine weight / bias, in a first-order training pass and in inference."""
    from style_big_gan_amd.train_parts import generators
    torch.manual_seed(1)
    syn = generators.SynthesisNetwork(w_dim=64, img_resolution=32, img_channels=3, channel_base=1024, channel_max=64, num_fp16_res=nfp,
                                      block_kwargs=generators.Synthblockkwargs(conv_clamp=256)).to(dev)
    for p in syn.parameters():
        if p.ndim == 0:
            torch.nn.init.constant_(p, 0.3)        # noise strengths off zero
    ws = torch.randn(6, syn.num_ws, 64, device=dev, requires_grad=True)
    results = []
    for on in (False, True):
        generators.style_bank_enabled = on
        try:
            syn.zero_grad(set_to_none=True); ws.grad = None
            torch.manual_seed(7)
            img = syn(ws, noise_mode='const')
            (img * torch.linspace(-1, 1, img.numel(), device=dev).view_as(img)).sum().backward()
            grads = {n: p.grad.clone() for n, p in syn.named_parameters() if 'affine' in n}
            with torch.no_grad():
                img_inf = syn(ws.detach(), noise_mode='const')
            results.append((img.detach(), ws.grad.clone(), grads, img_inf))
        finally:
            generators.style_bank_enabled = True
    (i0, g0, p0, f0), (i1, g1, p1, f1) = results
    def close(a, b, what):
        # fp32 network: the two fp32 summation orders; with bf16 blocks downstream a last-bit difference of a style can move a bf16 rounding
        tol = (2e-4 if nfp == 0 else 3e-2) * max(1e-3, float(a.abs().max()))
        assert float((a - b).abs().max()) <= tol, (what, float((a - b).abs().max()), tol)
    close(i0, i1, 'image'); close(f0, f1, 'inference image'); close(g0, g1, 'd ws')
    assert len(p0) == len(p1) >= 2 * 8
    for n in p0:
        close(p0[n], p1[n], n)
