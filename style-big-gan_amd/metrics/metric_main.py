"""Metric registry and reporting.  Interface of the reference's ``stylegan2ada/metrics/metric_main.py``: ``register_metric``,
``is_valid_metric``, ``list_valid_metrics``, ``calc_metric(metric, dataset_name, **MetricOptions kwargs)`` (:39-61: compute, broadcast
rank 0's numbers, wrap with timing metadata) and ``report_metric(result_dict, run_dir, snapshot_pkl)`` (:65-76: one JSON line to stdout
and to ``metric-<name>.jsonl``).  Metric names and their sample counts are the reference's (:81-150); the perceptual-path-length family
needs an LPIPS network and is not registered."""
import json
import os
import time

import torch

from ..utils import EasyDict
from . import metric_utils, scores

_metric_dict = dict()       # name -> fn(opts, dataset_name)


def register_metric(fn):
    assert callable(fn)
    _metric_dict[fn.__name__] = fn
    return fn


def is_valid_metric(metric):
    return metric in _metric_dict


def list_valid_metrics():
    return list(_metric_dict.keys())


def calc_metric(metric, dataset_name='image_folder', **kwargs):
    assert is_valid_metric(metric), f'unknown metric {metric}; known: {list_valid_metrics()}'
    opts = metric_utils.MetricOptions(**kwargs)
    start = time.time()
    results = _metric_dict[metric](opts, dataset_name=dataset_name)
    total = time.time() - start
    for key, value in list(results.items()):       # rank 0 holds the numbers; everybody gets them
        if opts.num_gpus > 1:
            t = torch.as_tensor(value, dtype=torch.float64, device=opts.device)
            torch.distributed.broadcast(tensor=t, src=0)
            value = float(t.cpu())
        results[key] = value
    m, s = divmod(int(round(total)), 60)
    return EasyDict(results=EasyDict(results), metric=metric, total_time=total, total_time_str=f'{m}m {s:02d}s' if m else f'{s}s', num_gpus=opts.num_gpus)


def report_metric(result_dict, run_dir=None, snapshot_pkl=None):
    metric = result_dict['metric']
    assert is_valid_metric(metric)
    if run_dir is not None and snapshot_pkl is not None:
        snapshot_pkl = os.path.relpath(snapshot_pkl, run_dir)
    line = json.dumps(dict(result_dict, snapshot_pkl=snapshot_pkl, timestamp=time.time()))
    print(line)
    if run_dir is not None and os.path.isdir(run_dir):
        with open(os.path.join(run_dir, f'metric-{metric}.jsonl'), 'at') as f:
            f.write(line + '\n')


def _full_dataset(opts, keep_flips=False):
    opts.dataset_kwargs.update(max_size=None) if keep_flips else opts.dataset_kwargs.update(max_size=None, xflip=False)


# -- primary metrics (reference :81-109)
@register_metric
def fid50k_full(opts, dataset_name='image_folder'):
    _full_dataset(opts)
    return dict(fid50k_full=scores.compute_fid(opts, dataset_name=dataset_name, max_real=None, num_gen=50000))


@register_metric
def kid50k_full(opts, dataset_name='image_folder'):
    _full_dataset(opts)
    return dict(kid50k_full=scores.compute_kid(opts, dataset_name=dataset_name, max_real=1000000, num_gen=50000, num_subsets=100, max_subset_size=1000))


@register_metric
def pr50k3_full(opts, dataset_name='image_folder'):
    _full_dataset(opts)
    precision, recall = scores.compute_pr(opts, dataset_name=dataset_name, max_real=200000, num_gen=50000, nhood_size=3, row_batch_size=10000, col_batch_size=10000)
    return dict(pr50k3_full_precision=precision, pr50k3_full_recall=recall)


@register_metric
def is50k(opts, dataset_name='image_folder'):
    _full_dataset(opts)
    mean, std = scores.compute_is(opts, dataset_name=dataset_name, num_gen=50000, num_splits=10)
    return dict(is50k_mean=mean, is50k_std=std)


# -- legacy metrics (reference :114-130): 50k reals, data-set flips kept
@register_metric
def fid50k(opts, dataset_name='image_folder'):
    _full_dataset(opts, keep_flips=True)
    return dict(fid50k=scores.compute_fid(opts, dataset_name=dataset_name, max_real=50000, num_gen=50000))


@register_metric
def kid50k(opts, dataset_name='image_folder'):
    _full_dataset(opts, keep_flips=True)
    return dict(kid50k=scores.compute_kid(opts, dataset_name=dataset_name, max_real=50000, num_gen=50000, num_subsets=100, max_subset_size=1000))


@register_metric
def pr50k3(opts, dataset_name='image_folder'):
    _full_dataset(opts, keep_flips=True)
    precision, recall = scores.compute_pr(opts, dataset_name=dataset_name, max_real=50000, num_gen=50000, nhood_size=3, row_batch_size=10000, col_batch_size=10000)
    return dict(pr50k3_precision=precision, pr50k3_recall=recall)
