"""The LiDAR thread of the camera-LiDAR-inertial configuration for a batch of sequences (BASELINE configs[3]; LidarInertialProcess,
SF/include/lidar_front_end/LidarFrontEnd.cpp:615-785) through tc2li_lidar_inertial_frontend_batch:
 - the time sort of UndistortPcl on the device reproduces the permutation std::sort leaves (libstdc++'s introsort: ties are not kept in
   input order, and the voxel filter sums in point order) -- checked alone against the oracle's std::sort on scans, ties, sorted /
   reversed / constant / organ-pipe inputs and short inputs, and through the host fallback a depth limit forces;
 - the batch against the one-scan entry points called in the reference's order (bit-identical: same kernel bodies, same host steps)
   and against the oracle (tolerances of tests/test_eskf.py);
 - map_incremental for all the batch's maps afterwards."""
import os

import numpy as np
import pytest

from test_eskf import mkstate
from test_undistort_gpu import imu_stream, lidar_state24

pytestmark = pytest.mark.gpu


def sorted_by_oracle(oracle, pts):
    """The permutation std::sort(points, time_list) leaves: the oracle's UndistortPcl with a single pose only sorts."""
    tagged = pts.copy()
    tagged["intensity"] = np.arange(len(pts), dtype=np.float32)
    ident24 = np.concatenate([np.eye(3).ravel(), np.zeros(3), np.eye(3).ravel(), np.zeros(3)])
    out = oracle.undistort(tagged, np.zeros((1, 22)), ident24)
    return out["intensity"].astype(np.int32)


def time_cases(pkg, oracle, synthetic):
    rng = np.random.default_rng(11)
    scan = oracle.lidar_preprocess(synthetic.lidar_scan(synthetic.Scene(2), 3))  # 64 beams share every time stamp
    cases = {"scan": scan["curvature"].copy()}
    n = 50000
    cases["random"] = rng.uniform(0, 100, n).astype(np.float32)
    cases["few values"] = rng.integers(0, 7, n).astype(np.float32)
    cases["constant"] = np.full(n, 3.5, np.float32)
    cases["sorted"] = np.sort(rng.uniform(0, 100, n)).astype(np.float32)
    cases["reversed"] = cases["sorted"][::-1].copy()
    cases["organ pipe"] = np.concatenate([np.arange(n // 2), np.arange(n // 2)[::-1]]).astype(np.float32)
    cases["sorted with ties"] = np.repeat(np.arange(n // 32), 32).astype(np.float32)
    cases["negative and zero"] = np.concatenate([rng.normal(0, 1, 1000), np.zeros(500), -np.zeros(500)]).astype(np.float32)
    for m in (1, 2, 3, 15, 16, 17, 18, 33, 64, 65, 1023, 1025):
        cases["n = %d" % m] = rng.integers(0, 5, m).astype(np.float32)
    return cases


def test_device_time_sort_is_std_sort(pkg, oracle, synthetic):
    fe = pkg.LidarFrontEnd(max_points_per_scan=70000, max_scans=1)
    limited = []
    for name, t in time_cases(pkg, oracle, synthetic).items():
        pts = np.zeros(len(t), pkg.capi.POINT_DTYPE)
        pts["curvature"] = t
        want = sorted_by_oracle(oracle, pts)
        got, hit_limit = fe.time_sort(pts)
        if hit_limit:  # std::sort itself leaves introsort for heap sort here; the batch entry sorts such a scan on the host
            limited.append(name)
            continue
        assert np.array_equal(got, want), name
        assert np.array_equal(np.sort(got), np.arange(len(t))), name
    assert set(limited) <= {"organ pipe"}, limited  # the classic median-of-three adversary
    # a depth limit the recursion must reach: flagged, not silently wrong
    pts = np.zeros(5000, pkg.capi.POINT_DTYPE)
    pts["curvature"] = np.random.default_rng(0).uniform(0, 1, 5000)
    _, hit_limit = fe.time_sort(pts, depth_limit=2)
    assert hit_limit
    perm, hit_limit = fe.time_sort(pts[:0])
    assert len(perm) == 0 and not hit_limit


def build_batch(pkg, oracle, synthetic, S, seed0=0):
    """S sequences: a raw scan, the map of the scan before it, IMU samples over the sweep, a perturbed filter state and covariance."""
    from scipy.spatial.transform import Rotation
    seqs = []
    for s in range(S):
        scene = synthetic.Scene(seed0 + s)
        frame = 1 + s
        raw = synthetic.lidar_scan(scene, frame)
        down0 = oracle.voxel_grid(oracle.lidar_preprocess(synthetic.lidar_scan(scene, frame - 1)))
        st0 = pkg.capi.pack_lidar_state(*synthetic.lidar_state(frame - 1)[:2])
        world0 = oracle.feature_extraction(oracle.KdTree(down0[:8]), down0, st0)["world"]
        R1, t1 = synthetic.sensor_pose(frame)
        rng = np.random.default_rng(100 + s)
        beg, end = 10.0 + 0.1 * s, 10.1 + 0.1 * s
        imu = imu_stream(beg - 0.012, end + 0.004, seed=s)
        imu[:, 1:4] = np.array([0.0, 0.0, 9.81]) + rng.normal(0, 0.02, (len(imu), 3))   # at rest in the sensor frame: the state is the scan's pose
        imu[:, 4:7] = rng.normal(0, 0.002, (len(imu), 3))
        x = mkstate(R1 @ Rotation.from_rotvec(rng.normal(0, 0.003, 3)).as_matrix(), t1 + rng.normal(0, 0.03, 3), g=(0, 0, -9.81))
        A = rng.normal(0, 1, (23, 23))
        P = A @ A.T * 1e-6 + np.diag([1e-3] * 3 + [1e-4] * 3 + [1e-5] * 6 + [1e-2] * 3 + [1e-5] * 6 + [1e-6] * 2)
        seqs.append(dict(raw=raw, world0=world0, imu=imu, x=x, P=P, times=[beg, end, beg - 0.001, 1.0]))
    return seqs


COV12 = np.array([0.1] * 3 + [0.1] * 3 + [1e-4] * 3 + [1e-4] * 3)


def run_batch(pkg, seqs, max_iter=3, **kw):
    import torch
    S = len(seqs)
    fe = pkg.LidarFrontEnd(max_points_per_scan=int(max(len(q["raw"]) for q in seqs)), max_scans=S)
    maps = []
    for q in seqs:
        m = pkg.LidarMap(); m.Build(q["world0"]); maps.append(m)
    raw = np.concatenate([q["raw"] for q in seqs])
    offs = np.concatenate([[0], np.cumsum([len(q["raw"]) for q in seqs])]).astype(np.int32)
    dev = torch.from_numpy(raw.view(np.uint8)).cuda()
    out = fe.inertial_frontend_batch(dev.data_ptr(), offs, maps, np.stack([q["x"] for q in seqs]), np.stack([q["P"] for q in seqs]),
                                     [q["imu"] for q in seqs], np.array([q["times"] for q in seqs]), COV12, max_iter=max_iter,
                                     stream=torch.cuda.current_stream().cuda_stream, **kw)
    return fe, maps, out


def run_one_by_one(pkg, q, max_iter=3, **kw):
    fe = pkg.LidarFrontEnd(max_points_per_scan=len(q["raw"]), max_scans=1)
    m = pkg.LidarMap(); m.Build(q["world0"])
    pts = fe.process(q["raw"])
    x, P, poses, last = pkg.capi.lidar_imu_propagate_cov(q["x"], q["P"], COV12, q["imu"], *q["times"], np.zeros(6))
    und = fe.undistort(pts, poses, lidar_state24(x))
    down = fe.voxel_filter(und)
    x2, P2, st = fe.eskf_update(m, down, x, P, max_iter=max_iter, **kw)
    return fe, m, dict(x=x2, P=P2, st=st, n_pre=len(pts), n_down=len(down), last=last, down=down)


def same_stats(a, b):
    return (a.calls, a.searches, a.converged, a.finished, a.effct_feat_num) == (b.calls, b.searches, b.converged, b.finished, b.effct_feat_num)


@pytest.mark.parametrize("S,max_iter,ext", [(3, 3, False), (2, 4, True)])
def test_batch_equals_one_scan_at_a_time_and_the_oracle(pkg, oracle, synthetic, S, max_iter, ext):
    seqs = build_batch(pkg, oracle, synthetic, S)
    fe, maps, (xs, Ps, stats, n_pre, n_down, last) = run_batch(pkg, seqs, max_iter=max_iter, extrinsic_est_en=ext)
    for s, q in enumerate(seqs):
        fe1, m1, one = run_one_by_one(pkg, q, max_iter=max_iter, extrinsic_est_en=ext)
        assert (n_pre[s], n_down[s]) == (one["n_pre"], one["n_down"]) and n_down[s] > 3000
        assert same_stats(stats[s], one["st"]) and stats[s].finished and stats[s].effct_feat_num > 1000
        assert stats[s].res_mean_last == one["st"].res_mean_last
        assert np.array_equal(xs[s], one["x"]) and np.array_equal(Ps[s], one["P"]) and np.array_equal(last[s], one["last"])
        # the oracle, stage by stage in the reference's order
        pts = oracle.lidar_preprocess(q["raw"])
        xo, Po, poses = oracle.imu_propagate_cov(q["x"], q["P"], COV12, q["imu"], *q["times"], np.zeros(6))[:3]
        down = oracle.voxel_grid(oracle.undistort(pts, poses, lidar_state24(xo)))
        assert len(down) == n_down[s]
        # (the device's sin / cos in UndistortPcl may round a coordinate differently: tests/test_undistort_gpu.py; the centroids then differ
        # in the last bit, nothing else)
        assert np.array_equal(down["curvature"], one["down"]["curvature"]) or np.allclose(down["curvature"], one["down"]["curvature"], rtol=1e-6)
        want_x, want_P, info = oracle.eskf_update(xo, Po, oracle.KdTree(q["world0"]), down, max_iter=max_iter, extrinsic_est_en=ext)
        assert (stats[s].calls, stats[s].searches, stats[s].converged, bool(stats[s].finished)) == (info["calls"], info["searches"], info["converged"], info["finished"])
        assert abs(stats[s].effct_feat_num - info["effct_feat_num"]) <= 2
        scale = max(1.0, np.abs(want_x[:3]).max())
        assert np.abs(xs[s][:3] - want_x[:3]).max() / scale < 1e-4
        assert np.abs(xs[s][3:12] - want_x[3:12]).max() < 1e-6
        assert np.allclose(xs[s][12:24], want_x[12:24], rtol=1e-4, atol=1e-6)
        wP = want_P.reshape(23, 23)
        assert np.abs(Ps[s] - wP).max() <= 1e-6 * np.abs(wP).max()
    # map_incremental of the whole batch on what the update left in the handle, against the one-scan path
    st24 = np.stack([lidar_state24(x) for x in xs])
    n_add, n_noneed, sizes = pkg.capi.map_incremental_batch(fe, np.arange(S, dtype=np.int32), maps, st24)
    for s, q in enumerate(seqs):
        fe1, m1, one = run_one_by_one(pkg, q, max_iter=max_iter, extrinsic_est_en=ext)
        size1, add1, noneed1 = m1.map_incremental(fe1, 0, lidar_state24(one["x"]))
        assert (sizes[s], n_add[s], n_noneed[s]) == (size1, add1, noneed1) and n_add[s] + n_noneed[s] > 0


def test_batch_mixed_convergence_and_empty_effect(pkg, oracle, synthetic):
    """Scans that converge at different iterations, one far from its map (no effective point: the state stays), tight limits for all."""
    seqs = build_batch(pkg, oracle, synthetic, 3, seed0=4)
    seqs[1]["x"] = seqs[1]["x"].copy(); seqs[1]["x"][:3] += 500.0
    seqs[2]["x"] = seqs[2]["x"].copy(); seqs[2]["x"][:3] += [0.15, -0.1, 0.05]
    for lim in (None, np.full(23, 1e-12)):
        fe, maps, (xs, Ps, stats, n_pre, n_down, last) = run_batch(pkg, seqs, max_iter=3, limit=lim)
        for s, q in enumerate(seqs):
            _, _, one = run_one_by_one(pkg, q, max_iter=3, limit=lim)
            assert same_stats(stats[s], one["st"]), (s, lim is None)
            assert np.array_equal(xs[s], one["x"]) and np.array_equal(Ps[s], one["P"])
        assert stats[1].effct_feat_num == 0 and not stats[1].finished
        if lim is not None:
            assert stats[0].calls == 4 and stats[0].converged == 0 and stats[0].searches == 2


def test_sort_fallback_gives_the_same_batch(pkg, oracle, synthetic):
    seqs = build_batch(pkg, oracle, synthetic, 2, seed0=7)
    _, _, ref = run_batch(pkg, seqs)
    os.environ["TC2LI_TEST_SORT_DEPTH"] = "4"   # every scan reaches the limit: the host sorts
    try:
        _, _, got = run_batch(pkg, seqs)
    finally:
        del os.environ["TC2LI_TEST_SORT_DEPTH"]
    assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1]) and np.array_equal(ref[4], got[4])


def test_batch_argument_checks(pkg, oracle, synthetic):
    seqs = build_batch(pkg, oracle, synthetic, 1)
    with pytest.raises(pkg.capi.Tc2liError):
        run_batch(pkg, seqs, R=0.0)
    q = dict(seqs[0], imu=np.zeros((70, 7)))
    with pytest.raises(pkg.capi.Tc2liError):
        run_batch(pkg, [q])


def test_scans_prepared_a_step_ahead_give_the_one_call_results(pkg, oracle, synthetic, monkeypatch):
    """tc2li_lidar_inertial_prepare_batch on a second handle (own stream) + the front end with dev_raw = NULL: the one-call results bit for bit,
    map_incremental included; a scan sent through the host sort (depth limit) too; NULL without prepared scans is refused."""
    import torch
    seqs = build_batch(pkg, oracle, synthetic, 3, seed0=7)
    fe0, maps0, (xs0, Ps0, stats0, n_pre0, n_down0, last0) = run_batch(pkg, seqs, max_iter=3)
    S = len(seqs)
    raw = np.concatenate([q["raw"] for q in seqs])
    offs = np.concatenate([[0], np.cumsum([len(q["raw"]) for q in seqs])]).astype(np.int32)
    dev = torch.from_numpy(raw.view(np.uint8)).cuda()
    args = (np.stack([q["x"] for q in seqs]), np.stack([q["P"] for q in seqs]), [q["imu"] for q in seqs], np.array([q["times"] for q in seqs]), COV12)
    for depth in (None, "3"):
        if depth:
            monkeypatch.setenv("TC2LI_TEST_SORT_DEPTH", depth)  # every scan reaches the limit: sorted on the host inside prepare
        fe = pkg.LidarFrontEnd(max_points_per_scan=int(max(len(q["raw"]) for q in seqs)), max_scans=S)
        maps = []
        for q in seqs:
            m = pkg.LidarMap(); m.Build(q["world0"]); maps.append(m)
        with pytest.raises(pkg.capi.Tc2liError):
            fe.inertial_frontend_batch(None, offs, maps, *args, max_iter=3)
        side = torch.cuda.Stream()
        assert fe.inertial_prepare_batch(dev.data_ptr(), offs, stream=side.cuda_stream) == S
        xs, Ps, stats, n_pre, n_down, last = fe.inertial_frontend_batch(None, offs, maps, *args, max_iter=3, stream=torch.cuda.current_stream().cuda_stream)
        assert np.array_equal(n_pre, n_pre0) and np.array_equal(n_down, n_down0)
        assert np.array_equal(xs, xs0) and np.array_equal(Ps, Ps0) and np.array_equal(last, last0)
        assert all(same_stats(a, b) and a.res_mean_last == b.res_mean_last for a, b in zip(stats, stats0))
        with pytest.raises(pkg.capi.Tc2liError):  # consumed
            fe.inertial_frontend_batch(None, offs, maps, *args, max_iter=3)
        st24 = np.stack([lidar_state24(x) for x in xs])
        got = pkg.capi.map_incremental_batch(fe, np.arange(S, dtype=np.int32), maps, st24)
        if depth is None:
            want = pkg.capi.map_incremental_batch(fe0, np.arange(S, dtype=np.int32), maps0, st24)
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
        assert all(np.array_equal(a.points(), b.points()) for a, b in zip(maps, maps0))


def test_a_batch_of_66_scans_takes_the_batch_forms_with_the_one_scan_results(pkg, oracle, synthetic):
    """64 scans and more switch the front end to its batch forms -- Preprocess::process in one pass per scan (k_pre_stream), the time sort's
    LDS stage with a few workgroups per scan that take the ranges in turn, the sorted voxel filter -- : every scan still gets the result of the
    one-scan entry points, bit for bit."""
    four = build_batch(pkg, oracle, synthetic, 4, seed0=11)
    seqs = [four[i % 4] for i in range(66)]
    fe, maps, (xs, Ps, stats, n_pre, n_down, last) = run_batch(pkg, seqs, max_iter=3)
    ones = [run_one_by_one(pkg, q, max_iter=3)[2] for q in four]
    for s in range(66):
        one = ones[s % 4]
        assert (n_pre[s], n_down[s]) == (one["n_pre"], one["n_down"]), s
        assert same_stats(stats[s], one["st"]) and stats[s].res_mean_last == one["st"].res_mean_last, s
        assert np.array_equal(xs[s], one["x"]) and np.array_equal(Ps[s], one["P"]) and np.array_equal(last[s], one["last"]), s
