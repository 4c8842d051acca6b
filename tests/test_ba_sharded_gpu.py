"""One local-BA window split over several ranks (tc2li_local_lv_bundle_adjustment_sharded; BASELINE configs[4]):
landmarks partitioned by l % world, the ranks' parts of the reduced camera system [S | b] summed by an all-reduce per LM trial.
The checker is the single-GPU entry, which the other BA tests hold against the oracle: world 1 over RCCL must agree bit for bit
(sums of one part), world 2 to rounding (the sum of two parts is not the single sum in order), with the same LM decisions."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POSE_RTOL = 1e-4  # BASELINE.json: 1e-4 relative on optimised SE3 poses; measured agreement is ~1e-12


def rel_pose_err(a, b):
    return max(np.abs(a[:4] - b[:4]).max(), np.abs(a[4:] - b[4:]).max() / max(1.0, np.abs(b[4:]).max()))


def _window(synthetic, seed, lidar):
    w = synthetic.ba_window(seed, n_opt=8, n_fix=10, n_points=1500, pose_noise=(0.1, 0.01))
    kw = {}
    if lidar:
        last = len(w["poses"]) - 1
        win = list(range(last, last - 4, -1))
        kw = dict(win_pose=win, clouds=synthetic.ba_window_clouds(w, win, n_points=2000), Tcl7=synthetic.TCL7, weight=1.0)
    return w, kw


@pytest.mark.parametrize("lidar", [False, True])
def test_world_of_one_over_rccl_is_the_single_gpu_result(pkg, synthetic, lidar):
    w, kw = _window(synthetic, 3, lidar)
    e = pkg.pack_ba_edges(w["edges"])
    if lidar:
        want = pkg.capi.local_lv_bundle_adjustment(w["poses"], w["fixed"], w["points"], e, w["cam"], kw["win_pose"], kw["clouds"], kw["Tcl7"], 1.0)
    else:
        want = pkg.local_bundle_adjustment(w["poses"], w["fixed"], w["points"], e, w["cam"])
    comm = pkg.capi.RcclComm(pkg.capi.RcclComm.unique_id(), 0, 1)
    try:
        got = pkg.capi.local_lv_bundle_adjustment_sharded(comm.shard(), w["poses"], w["fixed"], w["points"], e, w["cam"], **kw)
    finally:
        comm.close()
    assert got[4].iterations == want[4].iterations and got[4].trials == want[4].trials and got[4].iterations > 2
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3])
    assert got[4].final_chi2 == want[4].final_chi2


def _rank(rank, world, port, lidar, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import tc2li_loader
    pkg = tc2li_loader.load()
    from tc2li_slam_amd import synthetic
    torch.cuda.set_device(0)  # the box has one GPU: both ranks share it, the collective runs over gloo
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w, kw = _window(synthetic, 3, lidar)
        e = pkg.pack_ba_edges(w["edges"])
        shard, keep = pkg.capi.torch_allreduce_shard(rank, world)
        lo, eo = pkg.capi.ba_shard_select(e, len(w["points"]), rank, world)
        got = pkg.capi.local_lv_bundle_adjustment_sharded(shard, w["poses"], w["fixed"], w["points"], e, w["cam"], **kw)
        del keep
        out[rank] = dict(poses=got[0], points=got[1], chi2=got[2], depth=got[3], iterations=got[4].iterations, trials=got[4].trials,
                         final_chi2=got[4].final_chi2, initial_chi2=got[4].initial_chi2, owned=(int(lo.sum()), int(eo.sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lidar", [False, True])
def test_two_ranks(pkg, synthetic, lidar):
    import torch.multiprocessing as mp
    w, kw = _window(synthetic, 3, lidar)
    e = pkg.pack_ba_edges(w["edges"])
    if lidar:
        want = pkg.capi.local_lv_bundle_adjustment(w["poses"], w["fixed"], w["points"], e, w["cam"], kw["win_pose"], kw["clouds"], kw["Tcl7"], 1.0)
    else:
        want = pkg.local_bundle_adjustment(w["poses"], w["fixed"], w["points"], e, w["cam"])
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29900 + os.getpid() % 90
    mp.spawn(_rank, args=(2, port, lidar, out), nprocs=2, join=True)
    a, b = out[0], out[1]
    # every rank holds the whole, identical result
    for k in ("poses", "points", "chi2", "depth"):
        assert np.array_equal(a[k], b[k]), k
    assert a["owned"][0] + b["owned"][0] == len(w["points"]) and a["owned"][1] + b["owned"][1] == len(e)
    assert min(a["owned"]) > 0 and min(b["owned"]) > 0
    # ... which is the single-GPU result to rounding, reached by the same LM decisions
    assert a["iterations"] == b["iterations"] == want[4].iterations and a["trials"] == b["trials"] == want[4].trials
    assert abs(a["initial_chi2"] - want[4].initial_chi2) <= 1e-9 * want[4].initial_chi2
    assert abs(a["final_chi2"] - want[4].final_chi2) <= 1e-7 * want[4].final_chi2
    worst = max(rel_pose_err(a["poses"][k], want[0][k]) for k in range(len(want[0])))
    print("2 ranks vs 1 GPU: worst relative pose difference %.3g, chi2 %.6f vs %.6f" % (worst, a["final_chi2"], want[4].final_chi2))
    assert worst < POSE_RTOL and worst < 1e-7
    assert np.allclose(a["points"], want[1], rtol=1e-7, atol=1e-7)
    assert np.array_equal(a["depth"], want[3])
    assert np.allclose(a["chi2"], want[2], rtol=1e-5, atol=1e-7)


def test_shard_argument_errors(pkg, synthetic):
    w, _ = _window(synthetic, 0, False)
    e = pkg.pack_ba_edges(w["edges"])
    shard, keep = pkg.capi.torch_allreduce_shard(0, 1)
    bad = pkg.capi.BaShard(2, 2, shard.allreduce, None)  # rank out of range
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.local_lv_bundle_adjustment_sharded(bad, w["poses"], w["fixed"], w["points"], e, w["cam"])
    none = pkg.capi.BaShard(0, 1, None, None)  # no all-reduce
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.local_lv_bundle_adjustment_sharded(none, w["poses"], w["fixed"], w["points"], e, w["cam"])
    many = pkg.capi.BaShard(0, len(w["points"]) + 1, shard.allreduce, None)  # more ranks than landmarks
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.local_lv_bundle_adjustment_sharded(many, w["poses"], w["fixed"], w["points"], e, w["cam"])


def _rank_failing(rank, world, port, out, where="setup", timeout_s=None):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["TC2LI_TEST_SHARD_FAIL"] = "1:" + where  # rank 1 fails there (read by every rank, acted on by rank 1)
    import datetime
    import torch
    import torch.distributed as dist
    import tc2li_loader
    pkg = tc2li_loader.load()
    from tc2li_slam_amd import synthetic
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world, **({"timeout": datetime.timedelta(seconds=timeout_s)} if timeout_s else {}))
    try:
        w, kw = _window(synthetic, 3, False)
        e = pkg.pack_ba_edges(w["edges"])
        shard, keep = pkg.capi.torch_allreduce_shard(rank, world)
        try:
            pkg.capi.local_lv_bundle_adjustment_sharded(shard, w["poses"], w["fixed"], w["points"], e, w["cam"], **kw)
            out[rank] = "returned"
        except pkg.capi.Tc2liError as err:
            out[rank] = str(err)
        del keep
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_a_failing_rank_takes_the_others_with_it(pkg):
    """a rank that fails between two collectives must not leave the others blocked in the next all-reduce.  Rank 1 fails
    in its set-up; it joins rank 0's first collective with the status word set, rank 0 reads it and returns TC2LI_ERR_COMM."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29800 + os.getpid() % 90
    mp.spawn(_rank_failing, args=(2, port, out), nprocs=2, join=True)
    assert "injected failure" in out[1], out[1]
    assert "another rank failed" in out[0], out[0]


@pytest.mark.timeout(300)
def test_a_rank_failing_mid_trial_enters_no_collective(pkg):
    """a rank that fails where it cannot know the peers' next collective (after a trial's scalar sum: another trial, the next
    iteration or the result sum follows, decided by values it has not read) must not join any collective -- a sum of the wrong size would
    hang or corrupt the others.  It returns its own error at once; the healthy rank stays in its next all-reduce until the communicator's
    watchdog (here gloo's 8 s timeout) ends it, and returns TC2LI_ERR_COMM."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29700 + os.getpid() % 90
    mp.spawn(_rank_failing, args=(2, port, out, "trial", 8), nprocs=2, join=True)
    assert "injected failure" in out[1], out[1]
    assert "all-reduce callback returned" in out[0] or "another rank failed" in out[0], out[0]
