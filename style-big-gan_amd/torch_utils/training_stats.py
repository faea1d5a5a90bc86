"""Scalar training statistics accumulated on the device and reduced across ranks.

Counterpart of ``stylegan2ada/torch_utils/training_stats.py`` (``init_multiprocessing`` :34, ``report`` :56,
``report0`` :103, ``Collector`` :113, ``_sync`` :234-266): every reported tensor contributes the three moments
[count, sum, sum of squares] in float64; ``Collector.update()`` performs ONE all-reduce of the stacked moments over
RCCL (or gloo in CPU tests) and exposes per-name mean / std since the previous update.
"""
import re

import numpy as np
import torch

_MOMENTS = 3
_DTYPE = torch.float64

_rank = 0
_sync_device = None
_taps = dict()          # name -> [callables]: observers that see every reported tensor of that name (device-side consumers)
_pending = dict()       # name -> device -> float64[3] accumulated since the last _sync
_cumulative = dict()    # name -> float64[3] on CPU, all ranks, since start


def init_multiprocessing(rank, sync_device):
    """sync_device: device used for the cross-rank all-reduce, or None for single-process runs"""
    global _rank, _sync_device
    _rank, _sync_device = rank, sync_device


def add_tap(name, fn):
    """fn(tensor) is called with every non-empty tensor reported under `name` (detached, on its device) -- for consumers that must
    not wait for a Collector's host synchronisation (the ADA heuristic keeps its running sums on the device)."""
    _taps.setdefault(name, []).append(fn)
    return fn


def remove_tap(name, fn):
    if fn in _taps.get(name, []):
        _taps[name].remove(fn)


def report(name, value):
    """Accumulate `value` (scalar, tensor or array of any shape) under `name`; returns `value` unchanged."""
    slot = _pending.setdefault(name, dict())
    v = torch.as_tensor(value)
    if v.numel() == 0:
        return value
    for fn in _taps.get(name, ()):
        fn(v.detach())
    # The reference reduces every reported tensor to its three moments on the spot (eight launches of a few numbers each, ~20 reports per
    # iteration).  Here the (small) tensor is parked and the moments are formed once per name when a Collector asks for them.
    v = v.detach().flatten()
    dev = v.device
    entry = slot.get(dev)
    if entry is None:
        entry = slot[dev] = [None, []]          # [moments accumulated so far, parked tensors]
    if v.numel() <= 65536:
        entry[1].append(v.clone())
        if len(entry[1]) >= 64:
            _fold(entry)
    else:
        m = _moments(v)
        entry[0] = m if entry[0] is None else entry[0] + m
    return value


def _moments(v):
    v = v.to(_DTYPE)
    return torch.stack([torch.full([], float(v.numel()), dtype=_DTYPE, device=v.device), v.sum(), v.square().sum()])


def _fold(entry):
    """entry = [moments | None, parked tensors] -> moments of everything reported so far (None if nothing); empties the parking list"""
    parked, entry[1] = entry[1], []
    if parked:
        m = _moments(torch.cat([t.to(_DTYPE) for t in parked]) if len(parked) > 1 else parked[0])
        entry[0] = m if entry[0] is None else entry[0] + m
    return entry[0]


def report0(name, value):
    """report() on rank 0 only"""
    report(name, value if _rank == 0 else [])
    return value


def _sync(names):
    if not names:
        return []
    dev = _sync_device if _sync_device is not None else torch.device("cpu")
    rows = []
    for name in names:
        total = torch.zeros([_MOMENTS], dtype=_DTYPE, device=dev)
        for entry in _pending.get(name, {}).values():
            m = _fold(entry)
            if m is not None:
                total = total + m.to(dev)
        _pending[name] = dict()
        rows.append(total)
    delta = torch.stack(rows)
    if _sync_device is not None and torch.distributed.is_initialized():
        torch.distributed.all_reduce(delta)
    delta = delta.cpu()
    for i, name in enumerate(names):
        _cumulative[name] = _cumulative.get(name, torch.zeros([_MOMENTS], dtype=_DTYPE)) + delta[i]
    return [(name, _cumulative[name]) for name in names]


class Collector:
    """Mean / std of the statistics whose names match `regex`, measured between consecutive update() calls."""

    def __init__(self, regex=".*", keep_previous=True):
        self._regex = re.compile(regex)
        self._keep_previous = keep_previous
        self._cumulative = dict()
        self._moments = dict()
        self.update()
        self._moments.clear()

    def names(self):
        return [n for n in _pending if self._regex.fullmatch(n)]

    def update(self):
        if not self._keep_previous:
            self._moments.clear()
        for name, cum in _sync(self.names()):
            prev = self._cumulative.get(name, torch.zeros([_MOMENTS], dtype=_DTYPE))
            delta = cum - prev
            self._cumulative[name] = cum.clone()
            if float(delta[0]) != 0:
                self._moments[name] = delta

    def _get(self, name):
        assert self._regex.fullmatch(name)
        return self._moments.get(name, torch.zeros([_MOMENTS], dtype=_DTYPE))

    def num(self, name):
        return int(self._get(name)[0])

    def mean(self, name):
        d = self._get(name)
        return float("nan") if int(d[0]) == 0 else float(d[1] / d[0])

    def std(self, name):
        d = self._get(name)
        if int(d[0]) == 0 or not np.isfinite(float(d[1])):
            return float("nan")
        if int(d[0]) == 1:
            return 0.0
        mean, raw_var = float(d[1] / d[0]), float(d[2] / d[0])
        return float(np.sqrt(max(raw_var - np.square(mean), 0)))

    def as_dict(self):
        return {n: EasyStat(num=self.num(n), mean=self.mean(n), std=self.std(n)) for n in self.names()}

    def __getitem__(self, name):
        return self.mean(name)


class EasyStat(dict):
    __getattr__ = dict.__getitem__
