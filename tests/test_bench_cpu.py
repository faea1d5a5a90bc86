"""CPU: bench.py's own rank launcher (`python bench.py --gpus N` without torchrun) -- the counterpart of the reference's
starter.py:26-30 (one process per GPU).  The ranks rendezvous over gloo here (SBG_DIST_BACKEND=gloo), count themselves with an
all-reduce and rank 0 prints the JSON; with the RCCL backend and fewer devices than ranks the launcher must fail loudly instead
of printing a one-GPU number."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(kw)
    return env


def test_launcher_starts_n_ranks_over_gloo():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-only"], env=_env(SBG_DIST_BACKEND="gloo"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0's)"
    rec = json.loads(lines[0])
    assert rec == dict(ranks_seen=2, n_gpus=2, backend="gloo")


def test_launcher_refuses_more_ranks_than_devices():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a machine with fewer than two devices")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(SBG_DIST_BACKEND="nccl"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "2 ranks need 2 devices" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")], "no value may be printed"


def test_launcher_fails_when_a_rank_fails():
    # WORLD_SIZE is what the ranks check --gpus against: a launcher environment that disagrees makes every rank exit non-zero
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-only"], env=_env(SBG_DIST_BACKEND="gloo", WORLD_SIZE="3", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=3 but --gpus 2" in r.stderr
    sys.path.insert(0, ROOT)
    import bench
    old = dict(os.environ)
    try:
        os.environ["SBG_DIST_BACKEND"] = "gloo"
        assert bench.launch_ranks(2, ["--gpus", "2", "--workload", "no_such_workload"]) != 0      # argparse rejects it in every child
    finally:
        os.environ.clear(); os.environ.update(old)


def test_host_cores_is_bounded_by_affinity():
    sys.path.insert(0, ROOT)
    import bench
    n = bench.host_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0))
