"""GPU parity of the visual-inertial local BA with the oracle: same Levenberg-Marquardt
trials, optimised keyframe states within 1e-4 relative (BASELINE.json's bar for SE3 poses), per-edge chi2, outlier flags.
Both sides get the same pre-integrations (the product's, row a11), so that this test isolates the optimisation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def problem(pkg, oracle, synthetic, seed, **kw):
    w = synthetic.inertial_window(seed, **kw)
    pre, pre298 = [], []
    for s, t1, t2 in w["samples"]:
        p = pkg.capi.Preintegrated(w["bias6"], *synthetic.IMU_NOISE)
        p.preintegrate(s, t1, t2)
        pre.append(p)
        pre298.append(oracle.pack_preintegrated(p.fields(), w["bias6"]))
    w["pre"], w["pre298"] = pre, np.stack(pre298)
    return w


def rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


@pytest.mark.parametrize("seed,n_opt,n_pts,iters,lam", [(0, 6, 400, 10, 1.0), (1, 10, 900, 10, 1.0), (2, 8, 600, 4, 1e-2), (3, 3, 200, 10, 1.0)])
def test_local_inertial_ba(pkg, oracle, synthetic, seed, n_opt, n_pts, iters, lam):
    w = problem(pkg, oracle, synthetic, seed, n_opt=n_opt, n_points=n_pts)
    want = oracle.local_inertial_ba(w["kf33"], w["fixed"], w["has_imu"], w["calib24"], w["points"], w["edges"], w["link4"], w["pre298"], w["cam"],
                                    iterations=iters, lambda_init=lam)
    kf, pts, chi2, dpos, stats = pkg.capi.local_inertial_bundle_adjustment(w["kf33"], w["fixed"], w["has_imu"], w["calib24"], w["points"],
                                                                          pkg.pack_ba_edges(w["edges"]), w["link4"], w["pre"], w["cam"],
                                                                          iterations=iters, lambda_init=lam)
    assert stats.iterations == want[4]
    assert stats.trials == int(want[5]["trials"].sum())
    # (the product evaluates GetDeltaRotation / Velocity / Position in float like the reference -- NormalizeRotation as Eigen's float
    # JacobiSVD since round 4 -- the oracle's edges in double: the inertial chi2, weighted by ~1e6, differs in its sixth digit)
    assert abs(stats.initial_chi2 - want[6][0]) <= 3e-6 * want[6][0]
    assert abs(stats.final_chi2 - want[6][1]) <= 3e-6 * want[6][1]
    for k in range(len(kf)):
        assert rel(kf[k, :24], want[0][k, :24]) < RTOL          # camera and body pose
        assert np.allclose(kf[k, 24:], want[0][k, 24:], rtol=RTOL, atol=1e-6)  # velocity, biases
    assert np.array_equal(kf[0], w["kf33"][0])                  # the fixed keyframe
    assert np.allclose(pts, want[1], rtol=RTOL, atol=1e-5)
    assert np.array_equal(dpos, want[3])
    # the two sides evaluate the bias-corrected pre-integration in float / double (row a11): the optimum moves by ~1e-6, the
    # reprojection errors of single edges by ~1e-4 px
    assert np.allclose(chi2, want[2], rtol=5e-3, atol=1e-3)
    assert np.array_equal(chi2 > 7.815, want[2] > 7.815) or np.sum((chi2 > 7.815) != (want[2] > 7.815)) <= 1
    # it really optimises
    assert stats.final_chi2 < 0.3 * stats.initial_chi2
    err0 = np.linalg.norm(w["kf33"][:, 21:24] - w["kf33_true"][:, 21:24], axis=1).mean()
    err1 = np.linalg.norm(kf[:, 21:24] - w["kf33_true"][:, 21:24], axis=1).mean()
    assert err1 < 0.6 * err0


@pytest.mark.parametrize("seed,n_opt,weight", [(0, 6, 1.0), (1, 8, 100.0), (2, 7, 10.0)])
def test_local_lviba(pkg, oracle, synthetic, seed, n_opt, weight):
    """OptimizerWithLidar::LocalLVIBA: the inertial BA plus EdgeLidar on the first 6 optimisable keyframes (most recent first)."""
    w = problem(pkg, oracle, synthetic, seed, n_opt=n_opt, n_points=400)
    K = len(w["kf33"])
    win = list(range(K - 1, K - 7, -1))
    clouds = synthetic.inertial_window_clouds(w, win, n_points=2400, seed=seed)
    tbl = synthetic.tbl7()
    want = oracle.local_lviba(w["kf33"], w["fixed"], w["has_imu"], w["calib24"], w["points"], w["edges"], w["link4"], w["pre298"], w["cam"], win,
                              clouds, synthetic.TCL7, tbl, weight)
    kf, pts, chi2, dpos, stats, ls = pkg.capi.local_lvi_bundle_adjustment(w["kf33"], w["fixed"], w["has_imu"], w["calib24"], w["points"],
                                                                         pkg.pack_ba_edges(w["edges"]), w["link4"], w["pre"], w["cam"], win, clouds,
                                                                         synthetic.TCL7, tbl, weight)
    assert ls.n_planes == want[7] and ls.n_planes > 50
    assert stats.iterations == want[4]
    assert stats.trials == int(want[5]["trials"].sum())
    assert abs(stats.initial_chi2 - want[6][0]) <= 1e-6 * want[6][0]
    assert abs(stats.final_chi2 - want[6][1]) <= 1e-5 * want[6][1]
    # the poses pass through float in UpdatePose: the residual moves in steps of ~1e-6 relative
    assert abs(ls.residual - want[8]["error"]) <= 1e-4 * want[8]["error"]
    for k in range(len(kf)):
        assert rel(kf[k, :24], want[0][k, :24]) < RTOL
        assert np.allclose(kf[k, 24:], want[0][k, 24:], rtol=RTOL, atol=1e-5)
    assert np.allclose(pts, want[1], rtol=RTOL, atol=1e-4)
    assert np.array_equal(dpos, want[3])
    # the edge changes the result (it is really in the system)
    plain = pkg.capi.local_inertial_bundle_adjustment(w["kf33"], w["fixed"], w["has_imu"], w["calib24"], w["points"], pkg.pack_ba_edges(w["edges"]),
                                                      w["link4"], w["pre"], w["cam"])
    assert np.abs(plain[0][:, 21:24] - kf[:, 21:24]).max() > 1e-5


def test_lviba_argument_errors(pkg, oracle, synthetic):
    w = problem(pkg, oracle, synthetic, 0, n_opt=6, n_points=200)
    clouds = synthetic.inertial_window_clouds(w, [6, 5], n_points=300)
    args = (w["kf33"], w["fixed"], w["has_imu"], w["calib24"], w["points"], pkg.pack_ba_edges(w["edges"]), w["link4"], w["pre"], w["cam"])
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.local_lvi_bundle_adjustment(*args, [6, 99], clouds, synthetic.TCL7, synthetic.tbl7(), 1.0)
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.local_lvi_bundle_adjustment(*args, [6, 5], [clouds[0], clouds[1][:0]], synthetic.TCL7, synthetic.tbl7(), 1.0)


def test_inertial_ba_argument_errors(pkg, oracle, synthetic):
    w = problem(pkg, oracle, synthetic, 0, n_opt=3, n_points=150)
    bad = w["link4"].copy(); bad[0, 1] = 99
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.local_inertial_bundle_adjustment(w["kf33"], w["fixed"], w["has_imu"], w["calib24"], w["points"], pkg.pack_ba_edges(w["edges"]), bad,
                                                  w["pre"], w["cam"])
    no_imu = w["has_imu"].copy(); no_imu[2] = 0
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.local_inertial_bundle_adjustment(w["kf33"], w["fixed"], no_imu, w["calib24"], w["points"], pkg.pack_ba_edges(w["edges"]), w["link4"],
                                                  w["pre"], w["cam"])
    # without inertial links it is a visual BA in the ImuCamPose parameterisation
    r = pkg.capi.local_inertial_bundle_adjustment(w["kf33"], w["fixed"], w["has_imu"], w["calib24"], w["points"], pkg.pack_ba_edges(w["edges"]),
                                                  np.zeros((0, 4)), [], w["cam"])
    assert r[4].iterations >= 1 and r[4].final_chi2 < r[4].initial_chi2


def _lvi_window(pkg, oracle, synthetic, seed, n_opt, n_pts, lidar=True, cloud_points=2400, **kw):
    w = problem(pkg, oracle, synthetic, seed, n_opt=n_opt, n_points=n_pts)
    K = len(w["kf33"])
    d = dict(kf33=w["kf33"], fixed=w["fixed"], has_imu=w["has_imu"], points=w["points"], edges=pkg.pack_ba_edges(w["edges"]), link4=w["link4"],
             pre=w["pre"], **kw)
    if lidar:
        win = list(range(K - 1, K - 7, -1))
        d.update(win_kf=win, clouds=synthetic.inertial_window_clouds(w, win, n_points=cloud_points, seed=seed), Tcl7=synthetic.TCL7, Tbl7=synthetic.tbl7(),
                 weight=1.0 + seed)
    return w, d


def test_lviba_batch_equals_the_one_window_calls(pkg, oracle, synthetic):
    """tc2li_local_lvi_bundle_adjustment_batch (lock step): every window's states, points, per-edge chi2 and counters are those of
    tc2li_local_lvi_bundle_adjustment -- windows of different sizes, with and without the LiDAR edge, 4 and 10 iterations, lambda 1e-2 / 1."""
    specs = [(0, 6, 400, True, {}), (1, 10, 900, True, {}), (2, 7, 500, False, {}), (3, 8, 600, True, dict(iterations=4, lambda_init=1e-2)),
             (4, 10, 900, True, {}), (5, 3, 200, False, {})]
    wins = [_lvi_window(pkg, oracle, synthetic, s, n, p, lidar=l, **kw) for s, n, p, l, kw in specs]
    w0 = wins[0][0]
    batch = pkg.capi.LviBatch([d for _, d in wins], w0["calib24"], w0["cam"])
    for conc in (8, 1):  # lock step; one window after the other
        assert batch.run(max_concurrency=conc) == len(wins)
        for i, (w, d) in enumerate(wins):
            it, lam = d.get("iterations", 10), d.get("lambda_init", 1.0)
            if "win_kf" in d:
                kf, pts, chi2, dpos, st, ls = pkg.capi.local_lvi_bundle_adjustment(d["kf33"], d["fixed"], d["has_imu"], w["calib24"], d["points"], d["edges"], d["link4"],
                                                                                   d["pre"], w["cam"], d["win_kf"], d["clouds"], d["Tcl7"], d["Tbl7"], d["weight"],
                                                                                   iterations=it, lambda_init=lam)
            else:
                kf, pts, chi2, dpos, st = pkg.capi.local_inertial_bundle_adjustment(d["kf33"], d["fixed"], d["has_imu"], w["calib24"], d["points"], d["edges"],
                                                                                    d["link4"], d["pre"], w["cam"], iterations=it, lambda_init=lam)
                ls = None
            bkf, bpts, bchi2, bdpos, bst, bls = batch.result(i)
            assert batch.results[i] == st.iterations == bst.iterations and bst.trials == st.trials and bst.n_free_poses == st.n_free_poses, i
            assert bst.initial_chi2 == st.initial_chi2 and bst.final_chi2 == st.final_chi2 and bst.final_lambda == st.final_lambda, i
            assert np.array_equal(bkf, kf) and np.array_equal(bpts, pts) and np.array_equal(bchi2, chi2) and np.array_equal(bdpos, dpos), i
            if ls is not None:
                assert (bls.n_planes, bls.hessian_evaluations, bls.residual, bls.chi2) == (ls.n_planes, ls.hessian_evaluations, ls.residual, ls.chi2), i
            assert st.final_chi2 < st.initial_chi2


def test_lviba_batch_group_equals_the_batch_call(pkg, oracle, synthetic):
    """tc2li_local_lvi_bundle_adjustment_batch_group (one lock-step group on a context of the caller's choice, for several mapping workers):
    the windows' results are those of tc2li_local_lvi_bundle_adjustment_batch."""
    specs = [(6, 5, 300, True), (7, 8, 500, False), (8, 6, 400, True)]
    wins = [_lvi_window(pkg, oracle, synthetic, s, n, p, lidar=l) for s, n, p, l in specs]
    w0 = wins[0][0]
    batch = pkg.capi.LviBatch([d for _, d in wins], w0["calib24"], w0["cam"])
    assert batch.run(max_concurrency=8) == len(wins)
    ref = [tuple(np.copy(a) for a in batch.result(i)[:4]) + (batch.result(i)[4].trials, batch.result(i)[4].final_chi2) for i in range(len(wins))]
    for group in (0, 2):
        assert batch.run_group(group) == len(wins)
        for i in range(len(wins)):
            r = batch.result(i)
            assert all(np.array_equal(a, b) for a, b in zip(r[:4], ref[i][:4])) and (r[4].trials, r[4].final_chi2) == ref[i][4:], (group, i)


@pytest.mark.parametrize("form", ["full-width", "64x64-units"])  # the two block-sparse MFMA kernels of the dense windows' Schur product (round 5)
def test_lviba_large_window_and_benched_cloud(pkg, oracle, synthetic, monkeypatch, form):
    """The 25-keyframe `bLarge` window of LocalInertialBA / LocalLVIBA (opt_it 4, lambda 1e-2: Optimizer.cc:1516-1523, OptimizerWithLidar.cc:493-500,
    :618) against the oracle -- the reduced system has (6 + 9) * 25 unknowns and the MFMA Schur path is taken (k_ba_schur_full; k_ba_schur_units with
    TC2LI_BA_DENSE_FULL=0, the form of windows beyond 176 columns) -- and an LVIBA window with the LiDAR edge at the benched cloud size."""
    monkeypatch.setenv("TC2LI_BA_DENSE_FULL", "1" if form == "full-width" else "0")
    w = problem(pkg, oracle, synthetic, 11, n_opt=25, n_points=1500)
    K = len(w["kf33"])
    win = list(range(K - 1, K - 7, -1))
    clouds = synthetic.inertial_window_clouds(w, win, n_points=2400, seed=11)
    tbl = synthetic.tbl7()
    for lidar in (False, True):
        if lidar:
            want = oracle.local_lviba(w["kf33"], w["fixed"], w["has_imu"], w["calib24"], w["points"], w["edges"], w["link4"], w["pre298"], w["cam"], win,
                                      clouds, synthetic.TCL7, tbl, 1.0, iterations=4, lambda_init=1e-2)
            kf, pts, chi2, dpos, stats, ls = pkg.capi.local_lvi_bundle_adjustment(w["kf33"], w["fixed"], w["has_imu"], w["calib24"], w["points"],
                                                                                 pkg.pack_ba_edges(w["edges"]), w["link4"], w["pre"], w["cam"], win, clouds,
                                                                                 synthetic.TCL7, tbl, 1.0, iterations=4, lambda_init=1e-2)
            assert ls.n_planes == want[7] and ls.n_planes > 50
        else:
            want = oracle.local_inertial_ba(w["kf33"], w["fixed"], w["has_imu"], w["calib24"], w["points"], w["edges"], w["link4"], w["pre298"], w["cam"],
                                            iterations=4, lambda_init=1e-2)
            kf, pts, chi2, dpos, stats = pkg.capi.local_inertial_bundle_adjustment(w["kf33"], w["fixed"], w["has_imu"], w["calib24"], w["points"],
                                                                                  pkg.pack_ba_edges(w["edges"]), w["link4"], w["pre"], w["cam"],
                                                                                  iterations=4, lambda_init=1e-2)
        assert stats.n_free_poses == 25 and stats.iterations == want[4] == 4
        assert stats.trials == int(want[5]["trials"].sum())
        assert abs(stats.initial_chi2 - want[6][0]) <= 1e-6 * want[6][0]
        assert abs(stats.final_chi2 - want[6][1]) <= 1e-5 * want[6][1]
        for k in range(len(kf)):
            assert rel(kf[k, :24], want[0][k, :24]) < RTOL
            assert np.allclose(kf[k, 24:], want[0][k, 24:], rtol=RTOL, atol=1e-5)
        assert np.allclose(pts, want[1], rtol=RTOL, atol=1e-4)
        assert stats.final_chi2 < stats.initial_chi2


@pytest.mark.gpu
def test_reduced_system_on_the_device_and_on_the_host(pkg, oracle, synthetic, monkeypatch):
    """The reduced system of an inertial window -- (6 + 9) unknowns per keyframe, Optimizer.cc:1635-1638 -- is solved by k_lvi_solve* on the device
    (band of the velocity / bias unknowns eliminated four columns per step, pose block on the matrix unit) or, with TC2LI_LVI_DEVICE_SOLVE=0, by
    the host's envelope LDL^T (reduced_solve.hpp): the same optimisation up to rounding, at both window sizes of LocalInertialBA, alone and in a
    lock-step batch."""
    results = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("TC2LI_LVI_DEVICE_SOLVE", mode)
        out = []
        for seed, n_opt, n_pts, iters, lam in ((21, 25, 1200, 4, 1e-2), (22, 10, 700, 6, 1.0), (23, 3, 300, 5, 1.0)):
            w = problem(pkg, oracle, synthetic, seed, n_opt=n_opt, n_points=n_pts)
            kf, pts, chi2, dpos, stats = pkg.capi.local_inertial_bundle_adjustment(w["kf33"], w["fixed"], w["has_imu"], w["calib24"], w["points"],
                                                                                  pkg.pack_ba_edges(w["edges"]), w["link4"], w["pre"], w["cam"],
                                                                                  iterations=iters, lambda_init=lam)
            assert stats.final_chi2 < stats.initial_chi2
            out.append((kf, pts, stats.iterations, stats.trials, stats.final_chi2))
        results[mode] = out
    for (kf1, pts1, it1, tr1, c1), (kf0, pts0, it0, tr0, c0) in zip(results["1"], results["0"]):
        assert (it1, tr1) == (it0, tr0)
        assert abs(c1 - c0) <= 1e-9 * abs(c0)
        assert np.allclose(kf1, kf0, rtol=1e-9, atol=1e-10) and np.allclose(pts1, pts0, rtol=1e-9, atol=1e-9)


def test_a_window_the_device_solve_does_not_take(pkg, oracle, synthetic):
    """k_lvi_solve* eliminates the velocity / bias unknowns as a band: a window with an inertial edge between keyframes more than two places apart
    (here keyframes 1 and 6; nothing the reference's temporal window produces) is solved by the host's envelope LDL^T by itself, and a batch that
    holds such a window beside ordinary ones keeps the ordinary ones in lock step and hands THAT window to the one-window path (round 6; the whole
    group went back before): each result is the one-window call's, bit for bit."""
    w = problem(pkg, oracle, synthetic, 31, n_opt=6, n_points=400)
    link4 = np.vstack([w["link4"], [1.0, 6.0, 0.0, 1.0]])
    pre = list(w["pre"]) + [w["pre"][2]]
    pre298 = np.vstack([w["pre298"], w["pre298"][2:3]])
    want = oracle.local_inertial_ba(w["kf33"], w["fixed"], w["has_imu"], w["calib24"], w["points"], w["edges"], link4, pre298, w["cam"], iterations=6, lambda_init=1.0)
    edges = pkg.pack_ba_edges(w["edges"])
    kf, pts, chi2, dpos, stats = pkg.capi.local_inertial_bundle_adjustment(w["kf33"], w["fixed"], w["has_imu"], w["calib24"], w["points"], edges, link4, pre, w["cam"],
                                                                          iterations=6, lambda_init=1.0)
    assert stats.iterations == want[4] and stats.trials == int(want[5]["trials"].sum())
    assert abs(stats.final_chi2 - want[6][1]) <= 1e-5 * want[6][1]
    for k in range(len(kf)):
        assert rel(kf[k, :24], want[0][k, :24]) < RTOL
        assert np.allclose(kf[k, 24:], want[0][k, 24:], rtol=RTOL, atol=1e-5)
    w2 = problem(pkg, oracle, synthetic, 32, n_opt=5, n_points=300)
    batch = pkg.capi.LviBatch([dict(kf33=w2["kf33"], fixed=w2["fixed"], has_imu=w2["has_imu"], points=w2["points"], edges=pkg.pack_ba_edges(w2["edges"]), link4=w2["link4"],
                                    pre=w2["pre"], iterations=6),
                               dict(kf33=w["kf33"], fixed=w["fixed"], has_imu=w["has_imu"], points=w["points"], edges=edges, link4=link4, pre=pre, iterations=6)],
                              w["calib24"], w["cam"])
    assert batch.run(max_concurrency=8) == 2
    # the same with the odd window in the minority of ONE group of three (two ordinary windows stay in lock step)
    w3 = problem(pkg, oracle, synthetic, 33, n_opt=5, n_points=300)
    three = pkg.capi.LviBatch([dict(kf33=w2["kf33"], fixed=w2["fixed"], has_imu=w2["has_imu"], points=w2["points"], edges=pkg.pack_ba_edges(w2["edges"]), link4=w2["link4"],
                                    pre=w2["pre"], iterations=6),
                               dict(kf33=w["kf33"], fixed=w["fixed"], has_imu=w["has_imu"], points=w["points"], edges=edges, link4=link4, pre=pre, iterations=6),
                               dict(kf33=w3["kf33"], fixed=w3["fixed"], has_imu=w3["has_imu"], points=w3["points"], edges=pkg.pack_ba_edges(w3["edges"]), link4=w3["link4"],
                                    pre=w3["pre"], iterations=6)], w["calib24"], w["cam"])
    assert three.run_group(2) == 3
    for i in (0, 1):
        a, b = three.result(i), batch.result(i)
        assert a[4].trials == b[4].trials and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    kf3, pts3, _, _, st3 = pkg.capi.local_inertial_bundle_adjustment(w3["kf33"], w3["fixed"], w3["has_imu"], w3["calib24"], w3["points"], pkg.pack_ba_edges(w3["edges"]),
                                                                    w3["link4"], w3["pre"], w3["cam"], iterations=6, lambda_init=1.0)
    assert three.result(2)[4].trials == st3.trials and np.array_equal(three.result(2)[0], kf3) and np.array_equal(three.result(2)[1], pts3)
    bkf, bpts, bchi2, bdpos, bst, _ = batch.result(1)
    assert bst.iterations == stats.iterations and bst.trials == stats.trials and bst.final_chi2 == stats.final_chi2
    assert np.array_equal(bkf, kf) and np.array_equal(bpts, pts)
    kf2, pts2, _, _, st2 = pkg.capi.local_inertial_bundle_adjustment(w2["kf33"], w2["fixed"], w2["has_imu"], w2["calib24"], w2["points"], pkg.pack_ba_edges(w2["edges"]),
                                                                    w2["link4"], w2["pre"], w2["cam"], iterations=6, lambda_init=1.0)
    bkf2, bpts2, _, _, bst2, _ = batch.result(0)
    assert bst2.trials == st2.trials and np.array_equal(bkf2, kf2) and np.array_equal(bpts2, pts2)
