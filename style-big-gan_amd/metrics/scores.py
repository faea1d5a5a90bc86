"""The scores computed from feature statistics: Frechet distance (FID), kernel distance (KID), Inception score (IS) and improved
precision / recall (PR).  Arithmetic of the reference's ``stylegan2ada/metrics/{frechet_inception_distance, kernel_inception_distance,
inception_score, precision_recall}.py``; each function cites the lines it follows.  Detector names are the reference's file names; the
files themselves are never fetched (see metric_utils.get_feature_detector)."""
import numpy as np
import scipy.linalg
import torch

from . import metric_utils

INCEPTION = 'inception-2015-12-05.pt'      # reference: nvlabs-fi-cdn URL / './inception-2015-12-05.pt' (frechet_inception_distance.py:22-23)
VGG16 = 'vgg16.pt'                          # reference: precision_recall.py:37


def frechet_distance(mu_a, sigma_a, mu_b, sigma_b):
    """|mu_a - mu_b|^2 + tr(S_a + S_b - 2 (S_a S_b)^(1/2))   (frechet_inception_distance.py:42-44)"""
    m = np.square(mu_a - mu_b).sum()
    s, _ = scipy.linalg.sqrtm(np.dot(sigma_a, sigma_b), disp=False)
    return float(np.real(m + np.trace(sigma_a + sigma_b - s * 2)))


def compute_fid(opts, max_real, num_gen, dataset_name='image_folder'):
    kw = metric_utils.detector_call_kwargs(opts, INCEPTION, dict(return_features=True))
    mu_real, sigma_real = metric_utils.compute_feature_stats_for_dataset(
        opts=opts, detector_url=INCEPTION, detector_kwargs=kw, rel_lo=0, rel_hi=0, capture_mean_cov=True, max_items=max_real,
        dataset_name=dataset_name).get_mean_cov()
    mu_gen, sigma_gen = metric_utils.compute_feature_stats_for_generator(
        opts=opts, detector_url=INCEPTION, detector_kwargs=kw, rel_lo=0, rel_hi=1, capture_mean_cov=True, max_items=num_gen,
        dataset_name=dataset_name).get_mean_cov()
    if opts.rank != 0:
        return float('nan')
    return frechet_distance(mu_gen, sigma_gen, mu_real, sigma_real)


def kernel_distance(real_features, gen_features, num_subsets, max_subset_size, rng=np.random):
    """unbiased MMD^2 estimate with the cubic polynomial kernel (x.y / n + 1)^3, averaged over random subsets
    (kernel_inception_distance.py:32-43; subsets are drawn generated-first, then real, from `rng`)"""
    n = real_features.shape[1]
    m = min(min(real_features.shape[0], gen_features.shape[0]), max_subset_size)
    t = 0
    for _ in range(num_subsets):
        x = gen_features[rng.choice(gen_features.shape[0], m, replace=False)]
        y = real_features[rng.choice(real_features.shape[0], m, replace=False)]
        a = (x @ x.T / n + 1) ** 3 + (y @ y.T / n + 1) ** 3
        b = (x @ y.T / n + 1) ** 3
        t += (a.sum() - np.diag(a).sum()) / (m - 1) - b.sum() * 2 / m
    return float(t / num_subsets / m)


def compute_kid(opts, max_real, num_gen, num_subsets, max_subset_size, dataset_name='image_folder'):
    kw = metric_utils.detector_call_kwargs(opts, INCEPTION, dict(return_features=True))
    real = metric_utils.compute_feature_stats_for_dataset(opts=opts, dataset_name=dataset_name, detector_url=INCEPTION, detector_kwargs=kw,
                                                          rel_lo=0, rel_hi=0, capture_all=True, max_items=max_real).get_all()
    gen = metric_utils.compute_feature_stats_for_generator(opts=opts, dataset_name=dataset_name, detector_url=INCEPTION, detector_kwargs=kw,
                                                           rel_lo=0, rel_hi=1, capture_all=True, max_items=num_gen).get_all()
    if opts.rank != 0:
        return float('nan')
    return kernel_distance(real, gen, num_subsets, max_subset_size)


def inception_score(gen_probs, num_splits):
    """exp(E_x KL(p(y|x) || p(y))) per split -> (mean, std) over the splits (inception_score.py:30-36)"""
    num_gen = gen_probs.shape[0]
    scores = []
    for i in range(num_splits):
        part = gen_probs[i * num_gen // num_splits:(i + 1) * num_gen // num_splits]
        kl = part * (np.log(part) - np.log(np.mean(part, axis=0, keepdims=True)))
        scores.append(np.exp(np.mean(np.sum(kl, axis=1))))
    return float(np.mean(scores)), float(np.std(scores))


def compute_is(opts, num_gen, num_splits, dataset_name='image_folder'):
    kw = metric_utils.detector_call_kwargs(opts, INCEPTION, dict(no_output_bias=True))
    probs = metric_utils.compute_feature_stats_for_generator(opts=opts, dataset_name=dataset_name, detector_url=INCEPTION, detector_kwargs=kw,
                                                             capture_all=True, max_items=num_gen).get_all()
    if opts.rank != 0:
        return float('nan'), float('nan')
    return inception_score(probs, num_splits)


def pairwise_distances(row_features, col_features, num_gpus, rank, col_batch_size):
    """[rows, cols] Euclidean distances, the column batches dealt round-robin to the ranks and gathered on rank 0
    (precision_recall.py:16-29)"""
    assert 0 <= rank < num_gpus
    num_cols = col_features.shape[0]
    num_batches = ((num_cols - 1) // col_batch_size // num_gpus + 1) * num_gpus
    col_batches = torch.nn.functional.pad(col_features, [0, 0, 0, -num_cols % num_batches]).chunk(num_batches)
    out = []
    for col_batch in col_batches[rank::num_gpus]:
        dist = torch.cdist(row_features.unsqueeze(0), col_batch.unsqueeze(0))[0]
        if num_gpus > 1:
            parts = [torch.empty_like(dist) for _ in range(num_gpus)]
            torch.distributed.all_gather(parts, dist.contiguous())
        else:
            parts = [dist]
        if rank == 0:
            out.extend(p.cpu() for p in parts)
    return torch.cat(out, dim=1)[:, :num_cols] if rank == 0 else None


def precision_recall(real_features, gen_features, nhood_size, row_batch_size, col_batch_size, num_gpus=1, rank=0):
    """precision = share of generated features inside the union of k-NN balls of the real ones; recall = the converse
    (precision_recall.py:48-60).  Features come in as the caller's dtype (the reference casts to fp16 on the device)."""
    results = dict()
    for name, manifold, probes in [('precision', real_features, gen_features), ('recall', gen_features, real_features)]:
        kth = []
        for manifold_batch in manifold.split(row_batch_size):
            dist = pairwise_distances(manifold_batch, manifold, num_gpus, rank, col_batch_size)
            kth.append(dist.to(torch.float32).kthvalue(nhood_size + 1).values.to(manifold.dtype) if rank == 0 else None)
        kth = torch.cat(kth) if rank == 0 else None
        pred = []
        for probes_batch in probes.split(row_batch_size):
            dist = pairwise_distances(probes_batch, manifold, num_gpus, rank, col_batch_size)
            pred.append((dist <= kth).any(dim=1) if rank == 0 else None)
        results[name] = float(torch.cat(pred).to(torch.float32).mean()) if rank == 0 else float('nan')
    return results['precision'], results['recall']


def compute_pr(opts, max_real, num_gen, nhood_size, row_batch_size, col_batch_size, dataset_name='image_folder'):
    kw = metric_utils.detector_call_kwargs(opts, VGG16, dict(return_features=True))
    half = torch.float16 if torch.device(opts.device).type == 'cuda' else torch.float32
    real = metric_utils.compute_feature_stats_for_dataset(opts=opts, dataset_name=dataset_name, detector_url=VGG16, detector_kwargs=kw, rel_lo=0, rel_hi=0,
                                                          capture_all=True, max_items=max_real).get_all_torch().to(half).to(opts.device)
    gen = metric_utils.compute_feature_stats_for_generator(opts=opts, dataset_name=dataset_name, detector_url=VGG16, detector_kwargs=kw, rel_lo=0, rel_hi=1,
                                                           capture_all=True, max_items=num_gen).get_all_torch().to(half).to(opts.device)
    return precision_recall(real, gen, nhood_size, row_batch_size, col_batch_size, opts.num_gpus, opts.rank)
