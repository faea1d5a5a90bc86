"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (style-big-gan_amd/parallel.py GradReducer) and the
cross-rank statistics reduction.  Covers: constructor broadcast of rank 0's weights, flat-bucket gradient views, the
"latest forward decides" no_sync latch, all-reduce averaging on the last accumulation round only, overlap hooks firing per
bucket, nan_to_num on the flat buckets, and identical post-step weights on both ranks."""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, init_file, results):
    sys.path.insert(0, ROOT)
    import style_big_gan_amd  # noqa: F401
    from style_big_gan_amd.parallel import GradReducer
    from style_big_gan_amd.torch_utils import misc, training_stats
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                  # different init per rank: the constructor must broadcast rank 0's
        net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 4), torch.nn.Linear(4, 1))
        red = GradReducer(net, world_size=world, bucket_bytes=256)      # tiny buckets -> several buckets
        assert len(red._buckets) >= 2
        w0 = [p.detach().clone() for p in net.parameters()]
        gathered = [torch.zeros_like(w0[0]) for _ in range(world)]
        dist.all_gather(gathered, w0[0])
        assert torch.equal(gathered[0], gathered[1]), "weights not broadcast"
        for p in net.parameters():
            assert p.grad is not None and p.grad.data_ptr() != 0       # views into the flat buckets

        torch.manual_seed(7 + rank)
        xs = [torch.randn(5, 6) for _ in range(2)]                      # two accumulation rounds, different data per rank
        red.zero_grad()
        for i, x in enumerate(xs):
            with misc.ddp_sync(red, sync=(i == len(xs) - 1)):
                y = red(x)
            y.square().mean().backward()
        launched = [b.work is not None for b in red._buckets]
        assert any(launched), "no bucket was reduced from the backward hooks"
        red.finish()
        # reference: average over ranks of the summed per-round gradients, computed with plain autograd
        ref = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 4), torch.nn.Linear(4, 1))
        ref.load_state_dict({k: v for k, v in zip(ref.state_dict().keys(), w0)})
        for x in xs:
            ref(x).square().mean().backward()
        for p, q in zip(net.parameters(), ref.parameters()):
            g = q.grad.clone()
            dist.all_reduce(g)
            assert torch.allclose(p.grad, g / world, atol=1e-6), "all-reduced gradient mismatch"

        # no_sync round only: nothing is exchanged, gradients stay local
        red.zero_grad()
        with misc.ddp_sync(red, sync=False):
            y = red(xs[0])
        y.square().mean().backward()
        assert all(b.work is None for b in red._buckets)
        local = [p.grad.clone() for p in net.parameters()]
        red.finish(reduce=False)
        for p, l in zip(net.parameters(), local):
            assert torch.equal(p.grad, l)

        # a phase whose last backward ran under no_sync is still exchanged when it ends: finish() leaves the mean on every rank
        red.zero_grad()
        with misc.ddp_sync(red, sync=False):
            y = red(xs[0])
        y.square().mean().backward()
        red.finish()
        for p, l in zip(net.parameters(), local):
            g = l.clone()
            dist.all_reduce(g)
            assert torch.allclose(p.grad, g / world, atol=1e-6)

        # SEVERAL synchronising backwards between zero_grad() and finish() -- what the path-length and gradient-penalty regularisers
        # do (calc_reg gets sync=True on every accumulation round, reference losses_base.py:94,109): the result must be the mean over
        # ranks of the sum over rounds, and no backward may accumulate into a bucket that is still being reduced
        for n_rounds in (2, 3):
            red.zero_grad()
            xr = [torch.randn(5, 6) for _ in range(n_rounds)]
            for x in xr:
                with misc.ddp_sync(red, sync=True):
                    y = red(x)
                y.square().mean().backward()
            red.finish()
            for q in ref.parameters():
                q.grad = None
            for x in xr:
                ref(x).square().mean().backward()
            for p, q in zip(net.parameters(), ref.parameters()):
                g = q.grad.clone()
                dist.all_reduce(g)
                assert torch.allclose(p.grad, g / world, atol=1e-6), "repeated synchronising backwards: wrong gradient"
            both = [torch.zeros_like(red._buckets[0].flat) for _ in range(world)]
            dist.all_gather(both, red._buckets[0].flat)
            assert torch.equal(both[0], both[1]), "ranks hold different gradients after finish()"

        # TWO synchronising backwards behind ONE forward (two regularisers of a round differentiating the same logits), the first of
        # which reaches only part of the parameters: its half-marked buckets must not launch early in the next armed backward, and the
        # second backward must not accumulate into a bucket that the first one put on the wire (pre-accumulation hook settles it)
        red.zero_grad()
        with misc.ddp_sync(red, sync=True):
            h = red.module[1](red.module[0](xs[0]))         # the wrapper's forward arms; run the layers by hand to get at the middle
            y = red(xs[0])
        tail_only = red.module[3](red.module[2](h.detach())).square().mean()       # gradients for layers 2, 3 only
        tail_only.backward()
        y.square().mean().backward(retain_graph=True)       # all parameters; buckets of layers 2, 3 may be in flight from `tail_only`
        y.abs().mean().backward()                            # and again, no forward in between
        red.finish()
        for q in ref.parameters():
            q.grad = None
        hr = ref[1](ref[0](xs[0]))
        ref[3](ref[2](hr.detach())).square().mean().backward()
        yr = ref(xs[0])
        yr.square().mean().backward(retain_graph=True); yr.abs().mean().backward()
        for p, q in zip(net.parameters(), ref.parameters()):
            g = q.grad.clone()
            dist.all_reduce(g)
            assert torch.allclose(p.grad, g / world, atol=1e-6), "backwards sharing one forward: wrong gradient"

        # instrumentation bench.py switches on: stalls behind exchanges, non-finite gradient elements before nan_to_num
        red.timing, red.nonfinite = [], torch.zeros([], dtype=torch.int64)
        red.zero_grad()
        with misc.ddp_sync(red, sync=False):                # nothing on the wire until finish(): the buckets can be written by hand
            y = red(xs[1])
        y.square().mean().backward()
        if rank == 0:
            red._buckets[-1].flat[0] = float("nan")         # after the exchange every rank sees it
        red.finish()
        assert len(red.timing) >= 1 and red.exposed_wait_ms() >= 0.0 and red.timing == []
        assert int(red.nonfinite) >= 1
        red.timing, red.nonfinite = None, None

        # nan_to_num on the flat buckets
        red.zero_grad()
        red._buckets[0].flat[0] = float("nan"); red._buckets[0].flat[1] = float("inf")
        red.finish(reduce=False)
        assert float(red._buckets[0].flat[0]) == 0.0 and float(red._buckets[0].flat[1]) == 1e5

        # optimizer step on the reduced gradients keeps ranks identical
        opt = torch.optim.Adam(net.parameters(), lr=1e-2)
        red.zero_grad()
        y = red(xs[1]); y.square().mean().backward(); red.finish(); opt.step()
        w1 = torch.cat([p.detach().flatten() for p in net.parameters()])
        both = [torch.zeros_like(w1) for _ in range(world)]
        dist.all_gather(both, w1)
        assert torch.equal(both[0], both[1]), "ranks diverged after the step"
        misc.check_ddp_consistency(net)

        # training_stats: one all-reduce of the stacked moments
        training_stats.init_multiprocessing(rank, torch.device("cpu"))
        col = training_stats.Collector(regex="Loss/.*")
        training_stats.report("Loss/a", torch.tensor([1.0 + rank, 3.0 + rank]))
        col.update()
        assert abs(col.mean("Loss/a") - 2.5) < 1e-9 and col.num("Loss/a") == 4
        results[rank] = "ok"
    finally:
        dist.destroy_process_group()


def _engine_worker(rank, world, init_file, results):
    """StepEngine over two ranks: a 'Dboth' phase with the gradient penalty (its own synchronising D pass in EVERY accumulation round,
    after the adversarial pass -- the schedule that used to corrupt the flat buckets), two rounds per iteration, Adam + EMA; every
    rank must end with rank 0's G, D and G_ema (G_ema starts from the broadcast weights, not from the rank's own initialisation)."""
    sys.path.insert(0, ROOT)
    import style_big_gan_amd  # noqa: F401
    from style_big_gan_amd.torch_utils import misc
    from style_big_gan_amd.train_parts import trainers
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        eng = trainers.StepEngine("cpu", generator="cnn32_dcgan", discriminator="cnn32_dcgan", gen_kwargs=dict(z_dim=8, c_dim=0, img_resolution=32),
                                  disc_kwargs=dict(), loss_arch="base", loss="wasserstein", gen_regs=[], dis_regs=[("grad_pen", dict(alpha=10.))],
                                  optim_gen=("adam", dict(lr=1e-3, betas=[0.5, 0.9])), optim_disc=("adam", dict(lr=1e-3, betas=[0.5, 0.9])),
                                  g_reg_interval=0, d_reg_interval=0, batch=4, batch_gpu=2, ema_kimg=0.01, world_size=world, rank=rank, seed=3)
        assert [p.name for p in eng.phases] == ["Gboth", "Dboth"]
        misc.check_ddp_consistency(eng.G); misc.check_ddp_consistency(eng.D); misc.check_ddp_consistency(eng.G_ema)
        d0 = [p.detach().clone() for p in eng.D.parameters()]
        torch.manual_seed(50 + rank)                    # different reals and latents on each rank
        for _ in range(2):
            eng.train_iteration(torch.rand(4, 3, 32, 32) * 2 - 1, None)
        assert any(not torch.equal(a, b) for a, b in zip(d0, eng.D.parameters()))
        local_bn = r".*\.(running_mean|running_var|num_batches_tracked)"      # batch-norm statistics are per rank (broadcast_buffers=False, reference :592)
        for m in (eng.G, eng.D, eng.G_ema):
            misc.check_ddp_consistency(m, ignore_regex=local_bn)
        assert eng.cur_nimg == 2 * 4 * world
        results[rank] = "ok"
    finally:
        dist.destroy_process_group()


def _syncbn_worker(rank, world, init_file, results):
    """cross-replica batch norm over two ranks (reference sync_batchnorm/batchnorm.py:120-158: all-reduce of [sum x, sum x^2, n], mean /
    biased variance for the normalisation, UNBIASED variance into the running buffer; backward all-reduces the statistics' gradients)
    against one process normalising the concatenated batch with plain autograd: outputs, running statistics, gradients w.r.t. the rank's
    own slice of the input and w.r.t. gain / bias."""
    sys.path.insert(0, ROOT)
    import style_big_gan_amd  # noqa: F401
    from style_big_gan_amd.biggan import layers
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        torch.manual_seed(11)
        x_all = torch.randn(world * 3, 5, 4, 4, dtype=torch.float64).float() * 2 + 0.5      # both ranks draw the same full batch ...
        w_all = torch.randn(world * 3, 5, 4, 4)
        gain0, bias0 = torch.rand(5) + 0.5, torch.randn(5)
        sl = slice(rank * 3, (rank + 1) * 3)                                                  # ... and own one slice of it
        x = x_all[sl].clone().requires_grad_(True)
        mod = layers.bn(5, cross_replica=True).train()
        with torch.no_grad():
            mod.gain.copy_(gain0); mod.bias.copy_(bias0)
        y = mod(x)
        (y * w_all[sl]).sum().backward()

        # single-process statement over the whole batch
        xr = x_all.clone().requires_grad_(True)
        g, b = gain0.clone().requires_grad_(True), bias0.clone().requires_grad_(True)
        n = xr.numel() / xr.shape[1]
        mean = xr.sum([0, 2, 3]) / n
        sumvar = xr.square().sum([0, 2, 3]) - xr.sum([0, 2, 3]) * mean
        inv_std = torch.rsqrt(sumvar / n + mod.eps)
        yr = (xr - mean.view(1, -1, 1, 1)) * (inv_std * g).view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
        (yr * w_all).sum().backward()
        assert torch.allclose(y, yr[sl].detach(), atol=1e-5), "normalised output differs from the whole-batch statement"
        assert torch.allclose(x.grad, xr.grad[sl], atol=1e-5), "input gradient differs (the statistics' gradients must be all-reduced)"
        # gain / bias gradients are per-rank partial sums (the trainer's GradReducer averages them like every other parameter)
        for got, ref in ((mod.gain.grad, g.grad), (mod.bias.grad, b.grad)):
            tot = got.clone(); dist.all_reduce(tot)
            assert torch.allclose(tot, ref, atol=1e-4)
        assert torch.allclose(mod.stored_mean, 0.1 * mean.detach(), atol=1e-6)
        assert torch.allclose(mod.stored_var, 0.9 + 0.1 * (sumvar / (n - 1)).detach(), atol=1e-5)          # unbiased variance (:153,157)
        # both ranks hold the same running statistics
        both = [torch.zeros_like(mod.stored_var) for _ in range(world)]
        dist.all_gather(both, mod.stored_var)
        assert torch.equal(both[0], both[1])

        # class-conditional variant with per-sample gain / bias (what BigGAN's generator uses)
        cc = layers.ccbn(5, 7, torch.nn.Embedding, cross_replica=True).train()
        for p_ in cc.parameters():
            dist.broadcast(p_.detach(), src=0)
        labels = torch.tensor([1, 3, 3, 6, 0, 2])
        x2 = x_all[sl].clone().requires_grad_(True)
        y2 = cc(x2, labels[sl])
        (y2 * w_all[sl]).sum().backward()
        xr2 = x_all.clone().requires_grad_(True)
        gain = (1 + cc.gain(labels)).detach().view(-1, 5, 1, 1); bias = cc.bias(labels).detach().view(-1, 5, 1, 1)
        mean = xr2.sum([0, 2, 3]) / n
        inv_std = torch.rsqrt((xr2.square().sum([0, 2, 3]) - xr2.sum([0, 2, 3]) * mean) / n + cc.eps)
        yr2 = (xr2 - mean.view(1, -1, 1, 1)) * inv_std.view(1, -1, 1, 1) * gain + bias
        (yr2 * w_all).sum().backward()
        assert torch.allclose(y2, yr2[sl].detach(), atol=1e-5) and torch.allclose(x2.grad, xr2.grad[sl], atol=1e-5)
        results[rank] = "ok"
    finally:
        dist.destroy_process_group()


def _run_world(target, world=2):
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "rdzv")
        mgr = mp.Manager()
        results = mgr.dict()
        ctx = mp.get_context("spawn")
        procs = [ctx.Process(target=target, args=(r, world, init_file, results)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=180)
        for p in procs:
            assert p.exitcode == 0, f"worker exit code {p.exitcode}"
        assert dict(results) == {r: "ok" for r in range(world)}


def test_grad_reducer_world2_gloo():
    _run_world(_worker)


def test_step_engine_world2_repeated_sync_and_ema():
    _run_world(_engine_worker)


def test_cross_replica_batch_norm_world2():
    _run_world(_syncbn_worker)


def test_grad_reducer_single_process():
    sys.path.insert(0, ROOT)
    import style_big_gan_amd  # noqa: F401
    from style_big_gan_amd.parallel import GradReducer
    conv = torch.nn.Conv2d(4, 8, 3).to(memory_format=torch.channels_last)
    red = GradReducer(conv, world_size=1)
    assert conv.weight.grad.stride() == conv.weight.stride()          # channel-minor weights get channel-minor gradient views
    red.zero_grad()
    red(torch.randn(2, 4, 6, 6)).sum().backward()
    flat = red._buckets[0].flat
    assert float(flat.abs().sum()) > 0 and conv.weight.grad.data_ptr() >= flat.data_ptr()
    red.finish()
    assert red.grad_bytes() == sum(p.numel() for p in conv.parameters()) * 4
