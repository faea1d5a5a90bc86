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
        red._armed = False
        red.finish()
        for p, l in zip(net.parameters(), local):
            assert torch.equal(p.grad, l)

        # nan_to_num on the flat buckets
        red.zero_grad()
        red._buckets[0].flat[0] = float("nan"); red._buckets[0].flat[1] = float("inf")
        red._armed = False
        red.finish()
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


def test_grad_reducer_world2_gloo():
    world = 2
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "rdzv")
        mgr = mp.Manager()
        results = mgr.dict()
        ctx = mp.get_context("spawn")
        procs = [ctx.Process(target=_worker, args=(r, world, init_file, results)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=180)
        for p in procs:
            assert p.exitcode == 0, f"worker exit code {p.exitcode}"
        assert dict(results) == {0: "ok", 1: "ok"}


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
