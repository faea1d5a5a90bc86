"""Data parallelism for one process per GPU: bucketed gradient all-reduce over RCCL / xGMI, overlapped with backward.

The reference wraps G (or G.mapping / G.synthesis) and D in ``torch.nn.parallel.DistributedDataParallel`` with
``broadcast_buffers=False`` (train_parts/trainers.py:587-597, 883-893) and gates synchronisation per accumulation round with
``misc.ddp_sync`` -> ``module.no_sync()`` (torch_utils/misc.py:167-174).  ``GradReducer`` keeps that contract -- constructor
broadcast of rank 0's parameters and buffers, ``forward`` passthrough, ``no_sync()``, the "latest forward decides whether
the next backward synchronises" latch -- with an MI355X-oriented mechanism:

* every parameter's ``.grad`` is a persistent view into a few large flat fp32 buckets (default 32 MiB: xGMI is a set of
  point-to-point links, so fewer / larger collectives beat DDP's 25 MB default at these model sizes);
* a post-accumulate-grad hook marks parameters ready; the moment a bucket's last gradient of a synchronising backward is
  written, ONE asynchronous ``all_reduce`` of the whole bucket is enqueued on RCCL's stream, so the exchange of the layers
  that finish backward first overlaps the backward of the rest;
* ``finish()`` ends a phase: every bucket that still holds un-exchanged gradients is reduced, then waited for, averaged and
  passed through ``nan_to_num`` (the reference's per-parameter loop, trainers.py:745-747) over the flat buckets -- a handful
  of launches instead of hundreds;
* ``zero_grad()`` is one memset per bucket.

Several synchronising backwards between ``zero_grad()`` and ``finish()`` (path-length or gradient-penalty regularisers run one
per accumulation round; a phase with two regularisers runs two per round) are legal, as they are under DDP: a bucket keeps
the invariant ``flat = mean over ranks of what was exchanged so far + this rank's gradients since``.  A forward through the
wrapper first *settles* whatever an earlier backward left in flight (wait, scale by 1/world) so that the next backward never
accumulates into a buffer RCCL is still reducing; the next exchange then sums ``mean_so_far + local_new`` over the ranks and
the 1/world scaling gives ``mean_so_far + mean_new``.  A backward that arrives WITHOUT a forward in between (two regularisers
of one round differentiating the same logits) is covered by a tensor hook on every parameter, which autograd runs BEFORE it
accumulates the gradient: it settles that parameter's bucket if an exchange is in flight.  Readiness is per backward: arming a
forward clears what an earlier backward left half-marked (parameters it did not reach).

Instrumentation for ``bench.py`` (off by default, no host synchronisation when on): ``timing`` collects a device-event pair
around every wait for an exchange -- the time the compute stream is actually stalled by the collective, i.e. what backward did
NOT hide --, ``nonfinite`` accumulates the count of non-finite gradient elements per ``finish()`` BEFORE ``nan_to_num`` erases
them.

With ``world_size == 1`` nothing is communicated and the class is just the flat-gradient container.  The collective backend
is whatever ``torch.distributed`` was initialised with: ``nccl`` (= RCCL) on GPUs, ``gloo`` in the CPU tests.
"""
import contextlib
import time

import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("flat", "params", "ready", "work", "dirty")

    def __init__(self, flat, params):
        self.flat, self.params, self.ready, self.work = flat, params, set(), None
        self.dirty = False          # holds gradients of this rank that no exchange has covered yet


def _grad_view(flat, off, p):
    """view of flat[off : off + p.numel()] with p's shape AND p's memory layout (channel-minor weights keep a
    channel-minor gradient, so autograd's in-place accumulation and the optimizer see matching strides)"""
    seg = flat[off:off + p.numel()]
    if p.ndim == 4 and not p.is_contiguous() and p.is_contiguous(memory_format=torch.channels_last):
        o, i, kh, kw = p.shape
        return seg.view(o, kh, kw, i).permute(0, 3, 1, 2)
    return seg.view(p.shape)


class GradReducer(torch.nn.Module):
    def __init__(self, module, world_size=1, process_group=None, bucket_bytes=32 << 20, broadcast=True):
        super().__init__()
        self.module = module
        self.world_size = int(world_size)
        self.process_group = process_group
        self._sync_enabled = True       # toggled by no_sync()
        self._armed = True              # latched at forward(): does the next backward all-reduce?
        self._buckets = []
        self._bucket_of = dict()
        self.timing = None              # list of (event, event) / (t0, t1) pairs around waits when bench.py switches it on
        self.nonfinite = None           # device scalar: non-finite gradient elements seen by finish() before nan_to_num

        params = [p for p in module.parameters()]
        if self.world_size > 1 and broadcast:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.detach(), src=0, group=process_group)

        # buckets in reverse registration order (~ the order gradients become ready in backward)
        cur, cur_bytes = [], 0
        groups = []
        for p in reversed(params):
            nbytes = p.numel() * 4
            if cur and cur_bytes + nbytes > bucket_bytes:
                groups.append(cur); cur, cur_bytes = [], 0
            cur.append(p); cur_bytes += nbytes
        if cur:
            groups.append(cur)
        for grp in groups:
            total = sum(p.numel() for p in grp)
            flat = torch.zeros([total], dtype=grp[0].dtype, device=grp[0].device)
            b = _Bucket(flat, grp)
            off = 0
            for p in grp:
                assert p.dtype == flat.dtype and p.device == flat.device
                p.grad = _grad_view(flat, off, p)
                off += p.numel()
                self._bucket_of[id(p)] = b
                if self.world_size > 1:     # a single rank exchanges nothing: no per-parameter callbacks in its backward
                    was = p.requires_grad   # hooks can only be registered while the tensor requires grad
                    p.requires_grad_(True)
                    p.register_hook(lambda g, b=b: self._before_grad(b))         # runs before the gradient is accumulated
                    p.register_post_accumulate_grad_hook(self._on_grad)
                    p.requires_grad_(was)
            self._buckets.append(b)

    # -- module passthrough -----------------------------------------------------------------------------------------
    def forward(self, *args, **kwargs):
        self._armed = self._sync_enabled
        self._settle()              # an earlier backward of this phase may still be on the wire: complete it before the next one accumulates
        if self._armed:
            for b in self._buckets:     # readiness is per backward: forget parameters an earlier backward marked and this one may not reach
                b.ready.clear()
        return self.module(*args, **kwargs)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(super().__getattr__("module"), name)

    @contextlib.contextmanager
    def no_sync(self):
        old = self._sync_enabled
        self._sync_enabled = False
        try:
            yield
        finally:
            self._sync_enabled = old

    # -- gradient life cycle -----------------------------------------------------------------------------------------
    def zero_grad(self, set_to_none=False):
        for b in self._buckets:
            assert b.work is None, "GradReducer.zero_grad() with an all-reduce in flight: call finish() first"
            b.flat.zero_()
            b.ready.clear()
            b.dirty = False
            off = 0
            for p in b.params:          # re-install the views if something replaced them (e.g. optimizer.zero_grad(set_to_none=True))
                if p.grad is None or p.grad.data_ptr() != b.flat.data_ptr() + off * b.flat.element_size():
                    p.grad = _grad_view(b.flat, off, p)
                off += p.numel()

    def _launch(self, b):
        if b.work is None:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.process_group, async_op=True)
            b.dirty = False
            b.ready.clear()

    def _wait(self, b):
        """complete bucket b's exchange and turn its sum into a mean"""
        if self.timing is None:
            b.work.wait()
        elif b.flat.is_cuda:        # work.wait() makes the compute stream wait for RCCL's: the event pair brackets exactly that stall
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); b.work.wait(); e1.record()
            self.timing.append((e0, e1))
        else:
            t0 = time.perf_counter(); b.work.wait()
            self.timing.append((t0, time.perf_counter()))
        b.work = None
        b.flat.mul_(1.0 / self.world_size)

    def _settle(self):
        """wait for the exchanges in flight and turn their sums into means"""
        for b in self._buckets:
            if b.work is not None:
                self._wait(b)

    def _before_grad(self, b):
        """tensor hook of every parameter: autograd calls it before it accumulates into the bucket's view.  A backward that runs without
        a forward through the wrapper since the last synchronising backward finds the bucket on the wire: finish that exchange first."""
        if b.work is not None:
            self._wait(b)

    def _on_grad(self, p):
        if self.world_size <= 1:
            return
        b = self._bucket_of[id(p)]
        assert b.work is None, "GradReducer: gradient accumulated into a bucket whose all-reduce is in flight"
        b.dirty = True
        if not self._armed:
            return
        b.ready.add(id(p))
        if all((not q.requires_grad) or (id(q) in b.ready) for q in b.params):
            self._launch(b)

    def finish(self, nan_to_num=True, reduce=True):
        """End of a phase: exchange what no backward hook has exchanged yet (buckets whose parameters did not all receive a
        gradient, or whose last backward ran under ``no_sync``), wait, average, sanitise.  After this every rank holds the
        mean over ranks of everything accumulated since ``zero_grad()``.  ``reduce=False`` keeps un-exchanged gradients local."""
        if self.world_size > 1:
            if reduce:
                for b in self._buckets:
                    if b.dirty and b.work is None:
                        self._launch(b)
            self._settle()
        for b in self._buckets:
            b.ready.clear()
            if self.nonfinite is not None:
                self.nonfinite += (~torch.isfinite(b.flat)).sum()
            if nan_to_num:
                torch.nan_to_num(b.flat, nan=0, posinf=1e5, neginf=-1e5, out=b.flat)

    def exposed_wait_ms(self, reset=True):
        """sum of the stalls recorded in `timing` (ms); the caller synchronises the device first"""
        pairs, total = self.timing or [], 0.0
        for a, b in pairs:
            total += a.elapsed_time(b) if isinstance(a, torch.cuda.Event) else (b - a) * 1e3
        if reset and self.timing is not None:
            self.timing = []
        return total

    def grad_bytes(self):
        return sum(b.flat.numel() * b.flat.element_size() for b in self._buckets)
