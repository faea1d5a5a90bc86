"""Feature statistics for the quality metrics.

Counterpart of the reference's ``stylegan2ada/metrics/metric_utils.py``: ``MetricOptions`` (:22-33), ``get_feature_detector`` (:37-53),
``FeatureStats`` (:56-130), ``ProgressMonitor`` (:133-176), ``compute_feature_stats_for_dataset`` (:181-235: every rank takes items
``rank, rank + world, ...``, features are exchanged by per-rank broadcasts and interleaved back into data-set order, so every rank ends
with the statistics of the whole set) and ``compute_feature_stats_for_generator`` (:239-276).

MI355X-first differences:
* the running first / second moments are accumulated ON THE DEVICE in float64 (``x64.T @ x64`` is a small GEMM there) and only the final
  mean / covariance travel to the host; the reference copies every feature batch to the host and accumulates in numpy;
* the per-rank exchange is ONE ``all_gather`` instead of ``world`` broadcasts;
* nothing is fetched: detectors are local TorchScript files or callables, and cached statistics are ``.npz`` files (the reference pickles).
"""
import hashlib
import os
import time
import uuid

import numpy as np
import torch

from ..utils import EasyDict


class MetricOptions:
    """G + how to call it, the data set, the process layout, and where the feature detector comes from (there is no URL fetch in this build).
    ``detector`` overrides the detector a metric asks for: a callable (called as ``detector(images)``: it decides by itself what it returns, so
    the metric's ``detector_kwargs`` are NOT passed), the path of a local TorchScript file (called with the metric's ``detector_kwargs`` --
    ``return_features`` / ``no_output_bias`` -- like the reference's own files; accepted only for metrics whose expected detector has that
    file name, so that an Inception file is never used where VGG16 features are asked for), or a dict ``{detector file name or stem: path |
    callable}`` covering several metrics.  ``detector_dir``: a directory holding files under the reference's names."""

    def __init__(self, G=None, G_kwargs={}, dataset_kwargs={}, num_gpus=1, rank=0, device=None, progress=None, cache=True, detector=None,
                 detector_dir=None, cache_dir=None):
        assert 0 <= rank < num_gpus
        self.G = G
        self.G_kwargs = EasyDict(G_kwargs)
        self.dataset_kwargs = EasyDict(dataset_kwargs)
        self.num_gpus, self.rank = num_gpus, rank
        self.device = device if device is not None else torch.device('cuda', rank)
        self.progress = progress.sub() if progress is not None and rank == 0 else ProgressMonitor()
        self.cache = cache
        self.detector, self.detector_dir = detector, detector_dir
        self.cache_dir = cache_dir


# ------------------------------------------------------------------------------------------------------------------------------

_feature_detector_cache = dict()


def get_feature_detector_name(url):
    return os.path.splitext(str(url).split('/')[-1])[0]


def get_feature_detector(url, device=torch.device('cpu'), num_gpus=1, rank=0, verbose=False, detector_dir=None):
    """`url`: a callable (returned as it is), or the path / file name of a local TorchScript detector (looked up in `detector_dir` when it
    is not a file by itself).  http(s) locations are refused: this build has no network path."""
    assert 0 <= rank < num_gpus
    if callable(url):
        return url
    key = (url, str(device))
    if key not in _feature_detector_cache:
        path = str(url)
        if path.startswith(('http://', 'https://')):
            local = os.path.join(detector_dir, path.split('/')[-1]) if detector_dir else None
            if local is None or not os.path.isfile(local):
                raise RuntimeError(f'feature detector {path}: not fetched (no network path in this build). Place the TorchScript file locally and '
                                   'pass MetricOptions(detector=<path or callable>) or detector_dir=<directory holding it>.')
            path = local
        elif not os.path.isfile(path) and detector_dir and os.path.isfile(os.path.join(detector_dir, os.path.basename(path))):
            path = os.path.join(detector_dir, os.path.basename(path))
        if not os.path.isfile(path):
            raise RuntimeError(f'feature detector file {path} not found')
        _feature_detector_cache[key] = torch.jit.load(path, map_location=device).eval()
    return _feature_detector_cache[key]


# ------------------------------------------------------------------------------------------------------------------------------

class FeatureStats:
    def __init__(self, capture_all=False, capture_mean_cov=False, max_items=None):
        self.capture_all = capture_all
        self.capture_mean_cov = capture_mean_cov
        self.max_items = max_items
        self.num_items = 0
        self.num_features = None
        self.all_features = None        # list of float32 host arrays
        self.raw_mean = None            # float64 sums: numpy on the host, or a tensor on the device the features arrive on
        self.raw_cov = None

    def set_num_features(self, num_features):
        if self.num_features is not None:
            assert num_features == self.num_features
        else:
            self.num_features = num_features
            self.all_features = []
            self.raw_mean = np.zeros([num_features], dtype=np.float64)
            self.raw_cov = np.zeros([num_features, num_features], dtype=np.float64)

    def is_full(self):
        return (self.max_items is not None) and (self.num_items >= self.max_items)

    def _clip(self, n):
        """how many of `n` arriving items still fit under max_items"""
        if self.max_items is None:
            return n
        return max(min(n, self.max_items - self.num_items), 0)

    def append(self, x):
        x = np.asarray(x, dtype=np.float32)
        assert x.ndim == 2
        keep = self._clip(x.shape[0])
        if keep == 0:
            return
        x = x[:keep]
        self.set_num_features(x.shape[1])
        self.num_items += x.shape[0]
        if self.capture_all:
            self.all_features.append(x)
        if self.capture_mean_cov:
            x64 = x.astype(np.float64)
            self._host_moments()
            self.raw_mean += x64.sum(axis=0)
            self.raw_cov += x64.T @ x64

    def _host_moments(self):
        if torch.is_tensor(self.raw_mean):
            self.raw_mean, self.raw_cov = self.raw_mean.cpu().numpy(), self.raw_cov.cpu().numpy()

    def append_torch(self, x, num_gpus=1, rank=0):
        """features of this rank's share of a batch; with several ranks the shares are exchanged and interleaved (item i of the global
        batch came from rank i % world), so every rank accumulates the same, complete statistics (reference :99-108)"""
        assert isinstance(x, torch.Tensor) and x.ndim == 2
        assert 0 <= rank < num_gpus
        if num_gpus > 1:
            parts = [torch.empty_like(x) for _ in range(num_gpus)]
            torch.distributed.all_gather(parts, x.contiguous())
            x = torch.stack(parts, dim=1).flatten(0, 1)
        keep = self._clip(x.shape[0])
        if keep == 0:
            return
        x = x[:keep].to(torch.float32)
        self.set_num_features(x.shape[1])
        self.num_items += x.shape[0]
        if self.capture_all:
            self.all_features.append(x.cpu().numpy())
        if self.capture_mean_cov:
            if x.device.type == 'cpu':
                x64 = x.numpy().astype(np.float64)
                self._host_moments()
                self.raw_mean += x64.sum(axis=0)
                self.raw_cov += x64.T @ x64
            else:       # moments stay on the device: one [F, F] float64 GEMM per batch, nothing copied to the host
                if not torch.is_tensor(self.raw_mean):
                    self.raw_mean = torch.as_tensor(self.raw_mean, device=x.device)
                    self.raw_cov = torch.as_tensor(self.raw_cov, device=x.device)
                x64 = x.to(torch.float64)
                self.raw_mean += x64.sum(dim=0)
                self.raw_cov += x64.t() @ x64

    def get_all(self):
        assert self.capture_all
        return np.concatenate(self.all_features, axis=0)

    def get_all_torch(self):
        return torch.from_numpy(self.get_all())

    def get_mean_cov(self):
        assert self.capture_mean_cov
        self._host_moments()
        mean = self.raw_mean / self.num_items
        cov = self.raw_cov / self.num_items
        cov = cov - np.outer(mean, mean)
        return mean, cov

    # -- cache files: arrays + scalars in an .npz (loaded without unpickling anything)
    def save(self, path):
        self._host_moments()
        arrays = dict(capture_all=np.asarray(self.capture_all), capture_mean_cov=np.asarray(self.capture_mean_cov),
                      max_items=np.asarray(-1 if self.max_items is None else self.max_items), num_items=np.asarray(self.num_items),
                      num_features=np.asarray(-1 if self.num_features is None else self.num_features))
        if self.num_features is not None:
            arrays.update(raw_mean=self.raw_mean, raw_cov=self.raw_cov)
            if self.capture_all:
                arrays['all_features'] = self.get_all()
        with open(path, 'wb') as f:
            np.savez(f, **arrays)

    @staticmethod
    def load(path):
        with np.load(path, allow_pickle=False) as z:
            obj = FeatureStats(capture_all=bool(z['capture_all']), capture_mean_cov=bool(z['capture_mean_cov']),
                               max_items=None if int(z['max_items']) < 0 else int(z['max_items']))
            obj.num_items = int(z['num_items'])
            if int(z['num_features']) >= 0:
                obj.set_num_features(int(z['num_features']))
                obj.raw_mean, obj.raw_cov = z['raw_mean'].copy(), z['raw_cov'].copy()
                if obj.capture_all:
                    obj.all_features = [z['all_features'].copy()]
        return obj


# ------------------------------------------------------------------------------------------------------------------------------

class ProgressMonitor:
    """items-per-second reporting + mapping of a sub-task's progress onto a caller's [lo, hi] range (reference :133-176)"""

    def __init__(self, tag=None, num_items=None, flush_interval=1000, verbose=False, progress_fn=None, pfn_lo=0, pfn_hi=1000, pfn_total=1000):
        self.tag, self.num_items, self.verbose, self.flush_interval = tag, num_items, verbose, flush_interval
        self.progress_fn, self.pfn_lo, self.pfn_hi, self.pfn_total = progress_fn, pfn_lo, pfn_hi, pfn_total
        self.start_time = self.batch_time = time.time()
        self.batch_items = 0
        if self.progress_fn is not None:
            self.progress_fn(self.pfn_lo, self.pfn_total)

    def update(self, cur_items):
        assert (self.num_items is None) or (cur_items <= self.num_items)
        if (cur_items < self.batch_items + self.flush_interval) and (self.num_items is None or cur_items < self.num_items):
            return
        now = time.time()
        if self.verbose and self.tag is not None:
            per_item = (now - self.batch_time) / max(cur_items - self.batch_items, 1)
            print(f'{self.tag:<19s} items {cur_items:<7d} time {now - self.start_time:<10.1f}s ms/item {per_item * 1e3:.2f}')
        self.batch_time, self.batch_items = now, cur_items
        if (self.progress_fn is not None) and (self.num_items is not None):
            self.progress_fn(self.pfn_lo + (self.pfn_hi - self.pfn_lo) * (cur_items / self.num_items), self.pfn_total)

    def sub(self, tag=None, num_items=None, flush_interval=1000, rel_lo=0, rel_hi=1):
        span = self.pfn_hi - self.pfn_lo
        return ProgressMonitor(tag=tag, num_items=num_items, flush_interval=flush_interval, verbose=self.verbose, progress_fn=self.progress_fn,
                               pfn_lo=self.pfn_lo + span * rel_lo, pfn_hi=self.pfn_lo + span * rel_hi, pfn_total=self.pfn_total)


# ------------------------------------------------------------------------------------------------------------------------------

def detector_source(opts, detector_url):
    """what stands in for the detector `detector_url` (a reference file name) under `opts`: a callable or a path"""
    src = opts.detector
    if src is None:
        return detector_url
    if isinstance(src, dict):
        name = str(detector_url).split('/')[-1]
        for key in (name, get_feature_detector_name(name)):
            if key in src:
                return src[key]
        raise RuntimeError(f'no local detector for {name}: MetricOptions(detector=...) maps {sorted(src)}')
    if callable(src):
        return src
    want, have = get_feature_detector_name(detector_url), get_feature_detector_name(src)
    if want != have:
        raise RuntimeError(f'this metric computes on {want} features; the single detector file given is {os.path.basename(str(src))}. Pass a '
                           'directory (detector_dir) holding the reference file names, or a dict {name: path} (MetricOptions.detector)')
    return src


def detector_call_kwargs(opts, detector_url, detector_kwargs):
    """keyword arguments of the detector call: the metric's (return_features / no_output_bias: the reference's TorchScript files take them)
    unless a callable stands in -- it takes the images alone"""
    return {} if callable(detector_source(opts, detector_url)) else dict(detector_kwargs)


def _detector(opts, detector_url):
    return get_feature_detector(url=detector_source(opts, detector_url), device=opts.device, num_gpus=opts.num_gpus, rank=opts.rank,
                                detector_dir=opts.detector_dir)


def _as_rgb(images):
    return images.repeat([1, 3, 1, 1]) if images.shape[1] == 1 else images


def _cache_path(opts, dataset, detector_url, detector_kwargs, stats_kwargs):
    args = dict(dataset_kwargs=dict(opts.dataset_kwargs), detector_url=str(detector_url if not callable(detector_url) else getattr(detector_url, '__name__', 'callable')),
                detector_kwargs=detector_kwargs, stats_kwargs=stats_kwargs)
    md5 = hashlib.md5(repr(sorted(args.items())).encode('utf-8'))
    root = opts.cache_dir or os.path.join(os.environ.get('SBG_CACHE_DIR', os.path.join(os.path.expanduser('~'), '.cache', 'style_big_gan_amd')), 'gan-metrics')
    return os.path.join(root, f'{dataset.name}-{get_feature_detector_name(args["detector_url"])}-{md5.hexdigest()}.npz')


def compute_feature_stats_for_dataset(opts, detector_url, detector_kwargs, rel_lo=0, rel_hi=1, batch_size=64, data_loader_kwargs=None, max_items=None,
                                      dataset_name='image_folder', **stats_kwargs):
    from ..train_parts.datasets import datasets
    dataset = datasets[dataset_name](**opts.dataset_kwargs)
    if data_loader_kwargs is None:
        data_loader_kwargs = dict(pin_memory=(torch.device(opts.device).type == 'cuda'), num_workers=3, prefetch_factor=2)

    cache_file = None
    if opts.cache:      # all ranks must take the same branch: rank 0 looks, everybody hears
        cache_file = _cache_path(opts, dataset, detector_source(opts, detector_url), detector_kwargs, stats_kwargs)
        flag = os.path.isfile(cache_file) if opts.rank == 0 else False
        if opts.num_gpus > 1:
            t = torch.as_tensor(float(flag), dtype=torch.float32, device=opts.device)
            torch.distributed.broadcast(tensor=t, src=0)
            flag = float(t.cpu()) != 0
        if flag:
            return FeatureStats.load(cache_file)

    num_items = len(dataset)
    if max_items is not None:
        num_items = min(num_items, max_items)
    stats = FeatureStats(max_items=num_items, **stats_kwargs)
    progress = opts.progress.sub(tag='dataset features', num_items=num_items, rel_lo=rel_lo, rel_hi=rel_hi)
    detector = _detector(opts, detector_url)

    # rank r reads items r, r + world, ... (wrapping, so that all ranks run the same number of batches)
    item_subset = [(i * opts.num_gpus + opts.rank) % num_items for i in range((num_items - 1) // opts.num_gpus + 1)]
    for images, _labels in torch.utils.data.DataLoader(dataset=dataset, sampler=item_subset, batch_size=batch_size, **data_loader_kwargs):
        features = detector(_as_rgb(images.to(opts.device)), **detector_kwargs)
        stats.append_torch(features, num_gpus=opts.num_gpus, rank=opts.rank)
        progress.update(stats.num_items)

    if cache_file is not None and opts.rank == 0:
        os.makedirs(os.path.dirname(cache_file), exist_ok=True)
        temp_file = cache_file + '.' + uuid.uuid4().hex
        stats.save(temp_file)
        os.replace(temp_file, cache_file)       # atomic
    return stats


def compute_feature_stats_for_generator(opts, detector_url, detector_kwargs, rel_lo=0, rel_hi=1, batch_size=64, batch_gen=None, jit=False,
                                        dataset_name='image_folder', **stats_kwargs):
    """features of generated images: z ~ N(0, 1) on the device, labels drawn from the data set, images quantised to uint8 the way a saved
    PNG would be (reference :239-276).  `jit` is accepted for signature parity and ignored (there is no tracing compiler on this path)."""
    import copy
    from ..train_parts.datasets import datasets
    if batch_gen is None:
        batch_gen = min(batch_size, 4)
    assert batch_size % batch_gen == 0
    G = copy.deepcopy(opts.G).eval().requires_grad_(False).to(opts.device)
    dataset = datasets[dataset_name](**opts.dataset_kwargs)

    stats = FeatureStats(**stats_kwargs)
    assert stats.max_items is not None
    progress = opts.progress.sub(tag='generator features', num_items=stats.max_items, rel_lo=rel_lo, rel_hi=rel_hi)
    detector = _detector(opts, detector_url)
    c_dim = G.c_dim or 0
    with torch.no_grad():
        while not stats.is_full():
            images = []
            for _ in range(batch_size // batch_gen):
                z = torch.randn([batch_gen, G.z_dim], device=opts.device)
                c = np.stack([dataset.get_label(np.random.randint(len(dataset))) for _ in range(batch_gen)])
                c = torch.from_numpy(c).to(opts.device) if c_dim else torch.zeros([batch_gen, 0], device=opts.device)
                img = G(z, c, **opts.G_kwargs)
                images.append((img * 127.5 + 128).clamp(0, 255).to(torch.uint8))
            features = detector(_as_rgb(torch.cat(images)), **detector_kwargs)
            stats.append_torch(features, num_gpus=opts.num_gpus, rank=opts.rank)
            progress.update(stats.num_items)
    return stats
