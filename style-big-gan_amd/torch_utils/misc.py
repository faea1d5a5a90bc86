"""Host-side helpers shared by the models and the trainer.

Counterparts of the functions the hot path uses from ``stylegan2ada/torch_utils/misc.py``: ``assert_shape`` (:80),
``profiled_function`` (:98), ``InfiniteSampler`` (:109), ``params_and_buffers`` / ``copy_params_and_buffers``
(:145-163), ``ddp_sync`` (:167) and ``check_ddp_consistency`` (:179).  ``ddp_sync`` understands both
torch's DistributedDataParallel and this package's ``parallel.GradReducer`` wrapper.
"""
import contextlib
import re

import numpy as np
import torch


def assert_shape(tensor, ref_shape):
    if tensor.ndim != len(ref_shape):
        raise AssertionError(f"Wrong number of dimensions: got {tensor.ndim}, expected {len(ref_shape)}")
    for idx, (size, ref) in enumerate(zip(tensor.shape, ref_shape)):
        if ref is not None and int(size) != int(ref):
            raise AssertionError(f"Wrong size for dimension {idx}: got {size}, expected {ref}")


def profiled_function(fn):
    """run `fn` inside a record_function scope named after it (shows up as a roctx range under rocprofv3)"""
    def wrapper(*args, **kwargs):
        with torch.autograd.profiler.record_function(fn.__name__):
            return fn(*args, **kwargs)
    wrapper.__name__ = fn.__name__
    return wrapper


class InfiniteSampler(torch.utils.data.Sampler):
    """Endless shuffled index stream, sharded by rank, with windowed re-shuffling (reference misc.py:109-140)."""

    def __init__(self, dataset, rank=0, num_replicas=1, shuffle=True, seed=0, window_size=0.5):
        assert len(dataset) > 0 and num_replicas > 0 and 0 <= rank < num_replicas and 0 <= window_size <= 1
        super().__init__()
        self.dataset, self.rank, self.num_replicas = dataset, rank, num_replicas
        self.shuffle, self.seed, self.window_size = shuffle, seed, window_size

    def __iter__(self):
        order = np.arange(len(self.dataset))
        rnd, window = None, 0
        if self.shuffle:
            rnd = np.random.RandomState(self.seed)
            rnd.shuffle(order)
            window = int(np.rint(order.size * self.window_size))
        idx = 0
        while True:
            i = idx % order.size
            if idx % self.num_replicas == self.rank:
                yield order[i]
            if window >= 2:
                j = (i - rnd.randint(window)) % order.size
                order[i], order[j] = order[j], order[i]
            idx += 1


def params_and_buffers(module):
    assert isinstance(module, torch.nn.Module)
    return list(module.parameters()) + list(module.buffers())


def named_params_and_buffers(module):
    assert isinstance(module, torch.nn.Module)
    return list(module.named_parameters()) + list(module.named_buffers())


def copy_params_and_buffers(src_module, dst_module, require_all=False):
    src = dict(named_params_and_buffers(src_module))
    for name, tensor in named_params_and_buffers(dst_module):
        assert (name in src) or (not require_all)
        if name in src:
            tensor.copy_(src[name].detach()).requires_grad_(tensor.requires_grad)


@contextlib.contextmanager
def ddp_sync(module, sync):
    """Suppress the gradient all-reduce of a data-parallel wrapper unless `sync` (reference misc.py:167-174)."""
    assert isinstance(module, torch.nn.Module)
    if sync or not hasattr(module, "no_sync"):
        yield
    else:
        with module.no_sync():
            yield


def check_ddp_consistency(module, ignore_regex=None):
    """Assert that every parameter / buffer equals rank 0's copy (reference misc.py:179-188)."""
    assert isinstance(module, torch.nn.Module)
    for name, tensor in named_params_and_buffers(module):
        fullname = type(module).__name__ + "." + name
        if ignore_regex is not None and re.fullmatch(ignore_regex, fullname):
            continue
        tensor = tensor.detach()
        other = tensor.clone()
        torch.distributed.broadcast(tensor=other, src=0)
        assert (torch.nan_to_num(tensor) == torch.nan_to_num(other)).all(), fullname


class _Cat0(torch.autograd.Function):
    """torch.cat(tensors) along the batch axis as plain slice copies into one allocation.  aten's batched cat kernel moves a [128, 3, 256, 256]
    16-bit batch at ~125 GB/s on MI355X (815 us of a 111 ms step, profiles/r03b_kernel_stats.csv); a slice copy runs at the copy ceiling.
    Differentiable to any order (backward = views of the incoming gradient, themselves differentiable)."""

    @staticmethod
    def forward(ctx, *tensors):
        ctx.sizes = [t.shape[0] for t in tensors]
        first = tensors[0]
        fmt = torch.channels_last if (first.ndim == 4 and first.is_contiguous(memory_format=torch.channels_last) and not first.is_contiguous()) else torch.contiguous_format
        out = torch.empty([sum(ctx.sizes), *first.shape[1:]], dtype=first.dtype, device=first.device, memory_format=fmt)
        i = 0
        for t in tensors:
            out.narrow(0, i, t.shape[0]).copy_(t)
            i += t.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        return tuple(dout.split(ctx.sizes))


def cat0(tensors):
    """torch.cat(tensors, dim=0) for same-dtype tensors of equal trailing shape (see _Cat0); one tensor is returned as it is"""
    tensors = list(tensors)
    if len(tensors) == 1:
        return tensors[0]
    if any(t.dtype != tensors[0].dtype or t.shape[1:] != tensors[0].shape[1:] for t in tensors) or tensors[0].device.type != 'cuda':
        return torch.cat(tensors)
    return _Cat0.apply(*tensors)

