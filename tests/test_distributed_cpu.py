"""The N > 1 path of bench.py on CPU: world_size 2 over gloo (one process per rank, rendezvous on 127.0.0.1)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import time
    import tc2li_loader
    tc2li_loader.load()
    from tc2li_slam_amd import dist_util
    dist = dist_util.init("gloo", rank, world)
    # every rank owns its own sequences (no overlap, nothing lost)
    mine = dist_util.shard_units(10, rank, world)
    dist_util.barrier(dist)
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))      # rank 1 is the slow one
    dist_util.barrier(dist)
    elapsed = time.perf_counter() - t0 - (0.0 if rank else 0.0)
    local = 0.05 * (rank + 1)
    mx = dist_util.max_elapsed(dist, local)
    out[rank] = (mine, mx, dist_util.job_throughput(32, 5, world, mx), elapsed)
    dist.destroy_process_group()


def test_two_ranks_over_gloo():
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + os.getpid() % 400
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    (m0, mx0, thr0, e0), (m1, mx1, thr1, e1) = out[0], out[1]
    assert sorted(m0 + m1) == list(range(10)) and not set(m0) & set(m1)
    assert mx0 == mx1 == pytest.approx(0.10)                 # MAX over ranks, identical everywhere
    assert thr0 == thr1 == pytest.approx(32 * 5 * 2 / 0.10)   # whole-job units/s
    assert e0 >= 0.09 and e1 >= 0.09                          # the closing barrier holds the fast rank back


def test_single_rank_needs_no_process_group():
    sys.path.insert(0, ROOT)
    import tc2li_loader
    tc2li_loader.load()
    from tc2li_slam_amd import dist_util
    assert dist_util.init("gloo", 0, 1) is None
    assert dist_util.max_elapsed(None, 1.5) == 1.5
    assert dist_util.shard_units(5, 0, 1) == [0, 1, 2, 3, 4]
