"""`grid_sample` with arbitrarily high order gradients between input and output, as HIP kernels.

Same contract as the reference's ``stylegan2ada/torch_utils/ops/grid_sample_gradfix.py``: 2-D images, ``mode='bilinear'``,
``padding_mode='zeros'``, ``align_corners=False`` (:12-15).  The op is linear in ``input`` for a fixed grid, so

* forward            = gather                       (``sbg_grid_sample2d``),
* backward wrt input = scatter of ``grad_output``   (``sbg_grid_sample2d_bwd``; reference: ``aten::grid_sampler_2d_backward``, :63-64),
* its backward       = the forward applied to the incoming second-order gradient (reference :75-76),

and every level is an autograd Function again.  The gradient wrt ``grid`` exists at first order, like the reference's
(:52-53); differentiating THAT again is refused, as in the reference (:78).

``affine_grid_sample(input, theta, size)`` is ``grid_sample(input, F.affine_grid(theta, size, align_corners=False))``
(train_parts/augmentations.py:299-300) with the sampling positions generated inside the kernel -- the ``[N, H, W, 2]`` grid never
exists in HBM.  There is no CPU path.
"""
import torch

from ... import _lib

enabled = True      # kept for API parity with the reference module (trainers.py:513 sets it); the HIP op is always used


def _f32(t):
    return t if t.dtype == torch.float32 else t.to(torch.float32)


def _params(x_like, out_like, grid, theta):
    p = _lib.GridSampleParams()
    n, c, ih, iw = x_like.shape
    oh, ow = out_like.shape[2], out_like.shape[3]
    p.N, p.C, p.IH, p.IW, p.OH, p.OW = n, c, ih, iw, oh, ow
    p.xs_n, p.xs_c, p.xs_h, p.xs_w = x_like.stride()
    p.ys_n, p.ys_c, p.ys_h, p.ys_w = out_like.stride()
    if grid is not None:
        p.grid = grid.data_ptr()
    else:
        p.theta = theta.data_ptr()
    return p


def _check(input, grid, theta, size):
    _lib.require_cuda(input, "grid_sample")
    assert input.ndim == 4
    n = input.shape[0]
    if grid is not None:
        assert grid.ndim == 4 and grid.shape[0] == n and grid.shape[3] == 2
        return grid.shape[1], grid.shape[2]
    assert theta.shape == (n, 2, 3), f"theta must be [N, 2, 3], got {tuple(theta.shape)}"
    assert len(size) == 4 and size[0] == n and size[1] == input.shape[1]
    return int(size[2]), int(size[3])


def _positions(grid, theta):
    """dense fp32 copies of whichever tensor defines the sampling positions"""
    g = _f32(grid.detach()).contiguous() if grid is not None else None
    t = _f32(theta.detach()).contiguous() if theta is not None else None
    return g, t


def _gather(x, g, t, oh, ow):
    x = _f32(x)
    y = torch.empty([x.shape[0], x.shape[1], oh, ow], dtype=torch.float32, device=x.device)
    if y.numel():
        p = _params(x, y, g, t)
        p.x, p.y = x.data_ptr(), y.data_ptr()
        _lib.check(_lib.load().sbg_grid_sample2d(p, _lib.stream_ptr(x.device)), "sbg_grid_sample2d")
    return y


def _scatter(dy, x_shape, g, t, x=None, want_dx=True, theta_host=None):
    """-> (dx or None, dgrid or None); dgrid only when x is given"""
    lib = _lib.load()
    dy = _f32(dy)
    dx = torch.empty(x_shape, dtype=torch.float32, device=dy.device) if want_dx else None
    dgrid = torch.empty([dy.shape[0], dy.shape[2], dy.shape[3], 2], dtype=torch.float32, device=dy.device) if x is not None else None
    launch = dy.numel() and (want_dx or x is not None)
    overwrites = False
    if launch:
        ref = dx if dx is not None else x
        p = _params(ref, dy, g, t)
        p.dy = dy.data_ptr()
        if dx is not None:
            p.dx = dx.data_ptr()
        if x is not None:
            assert x.stride() == ref.stride()
            p.x, p.dgrid = x.data_ptr(), dgrid.data_ptr()
        if theta_host is not None and t is not None:
            assert theta_host.device.type == "cpu" and theta_host.dtype == torch.float32 and theta_host.is_contiguous() and theta_host.shape == t.shape
            p.theta_host = theta_host.data_ptr()
        overwrites = bool(lib.sbg_grid_sample2d_bwd_overwrites(p))      # deterministic gather kernel: writes every element of dx
    if dx is not None and not overwrites:
        dx.zero_()
    if launch:
        _lib.check(lib.sbg_grid_sample2d_bwd(p, _lib.stream_ptr(dy.device)), "sbg_grid_sample2d_bwd")
    return dx, dgrid


class _GridSampleForward(torch.autograd.Function):
    """(input, grid | None, theta | None, oh, ow, theta_host | None) -> output"""

    @staticmethod
    def forward(ctx, input, grid, theta, oh, ow, theta_host=None):
        g, t = _positions(grid, theta)
        y = _gather(input, g, t, oh, ow)
        ctx.save_for_backward(input if (grid is not None and grid.requires_grad) else None, g, t)
        ctx.x_shape, ctx.x_dtype = tuple(input.shape), input.dtype
        ctx.grid_dtype = grid.dtype if grid is not None else None
        ctx.theta_host = theta_host
        return y.to(input.dtype)

    @staticmethod
    def backward(ctx, grad_output):
        x, g, t = ctx.saved_tensors
        grad_input = grad_grid = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            grad_input, grad_grid = _GridSampleBackward.apply(grad_output, x, g, t, ctx.x_shape, ctx.needs_input_grad[0], ctx.theta_host)
            if grad_input is not None:
                grad_input = grad_input.to(ctx.x_dtype)
            if ctx.needs_input_grad[1]:
                grad_grid = grad_grid.to(ctx.grid_dtype)
            else:
                grad_grid = None
        if ctx.needs_input_grad[2]:
            raise RuntimeError("affine_grid_sample: the gradient with respect to theta is not implemented "
                               "(the augmentation pipe's transforms carry no gradient)")
        return grad_input, grad_grid, None, None, None, None


class _GridSampleBackward(torch.autograd.Function):
    """(grad_output, input | None, grid, theta, x_shape, want_dx, theta_host) -> (grad_input, grad_grid)"""

    @staticmethod
    def forward(ctx, grad_output, x, g, t, x_shape, want_dx, theta_host=None):
        xd = _f32(x.detach()).contiguous() if x is not None else None      # dense, so it shares dx's strides
        dx, dgrid = _scatter(grad_output, x_shape, g, t, x=xd, want_dx=want_dx or xd is None, theta_host=theta_host)
        ctx.save_for_backward(g, t)
        ctx.theta_host = theta_host
        ctx.oh, ctx.ow = grad_output.shape[2], grad_output.shape[3]
        ctx.mark_non_differentiable(*([dgrid] if dgrid is not None else []))
        return dx, dgrid

    @staticmethod
    def backward(ctx, grad2_grad_input, grad2_grad_grid):
        g, t = ctx.saved_tensors
        grad2_grad_output = None
        if ctx.needs_input_grad[0] and grad2_grad_input is not None:
            grad2_grad_output = _GridSampleForward.apply(grad2_grad_input, g, t, ctx.oh, ctx.ow, ctx.theta_host)
        # like the reference (:70-79): the second-order terms through `input` and `grid` (they exist only via grad_grid) are not produced
        return grad2_grad_output, None, None, None, None, None, None


def grid_sample(input, grid):
    """reference: grid_sample_gradfix.grid_sample(input, grid) (:24-27)"""
    oh, ow = _check(input, grid, None, None)
    return _GridSampleForward.apply(input, grid, None, oh, ow, None)


def affine_grid_sample(input, theta, size, theta_host=None):
    """grid_sample(input, affine_grid(theta, size, align_corners=False)) in one kernel; theta: [N, 2, 3] on the device.
    `theta_host`: the same values as a contiguous float32 CPU tensor, when the caller has them (the augmentation pipe composes its
    transforms on the host): the backward pass can then prove the per-pixel footprint small and run as a gather -- no atomics,
    no zero fill, bitwise reproducible -- instead of an atomic scatter."""
    oh, ow = _check(input, None, theta, size)
    return _GridSampleForward.apply(input, None, theta, oh, ow, theta_host)
