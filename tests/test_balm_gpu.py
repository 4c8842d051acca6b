"""GPU parity of the LiDAR plane term with the oracle: the edge alone (residual, Jacobian,
Hessian in the camera se3 parameterisation) and LocalLVBundleAdjustment with the edge in the Levenberg-Marquardt loop.
Bar (BASELINE.json): optimised SE3 poses within 1e-4 relative; the optimiser must also take the same trials."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POSE_RTOL = 1e-4


def rel_pose_err(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


@pytest.mark.parametrize("seed,W,n_pts", [(0, 4, 2400), (1, 6, 3000), (2, 2, 1500), (3, 12, 1200), (4, 20, 800)])
def test_lidar_edge_alone(pkg, oracle, synthetic, seed, W, n_pts):
    n_opt = max(6, W)
    w = synthetic.ba_window(seed, n_opt=n_opt, n_fix=4, n_points=200, pose_noise=(0.1, 0.01))
    last = len(w["poses"]) - 1
    win = list(range(last, last - W, -1))
    clouds = synthetic.ba_window_clouds(w, win, n_points=n_pts)
    n0, r0, J0, H0 = oracle.lidar_window_evaluate(w["poses"], win, clouds, synthetic.TCL7)
    n1, r1, J1, H1 = pkg.capi.lidar_window_evaluate(w["poses"], win, clouds, synthetic.TCL7)
    assert n1 == n0 and n0 > 10
    assert abs(r1 - r0) <= 1e-10 * abs(r0)
    assert np.allclose(J1, J0, rtol=1e-8, atol=1e-9 * np.abs(J0).max())
    assert np.allclose(H1, H0, rtol=1e-8, atol=1e-9 * np.abs(H0).max())
    # residual only (derivatives not requested)
    n2, r2, _, _ = pkg.capi.lidar_window_evaluate(w["poses"], win, clouds, synthetic.TCL7, derivatives=False)
    assert n2 == n0 and r2 == r1


@pytest.mark.parametrize("seed,weight,lam,W", [(0, 0.01, 0.0, 6), (0, 1.0, 0.0, 6), (1, 100.0, 0.0, 6), (2, 1.0, 100.0, 4), (3, 10.0, 0.0, 3)])
def test_local_lv_bundle_adjustment(pkg, oracle, synthetic, seed, weight, lam, W):
    w = synthetic.ba_window(seed, n_opt=12, n_fix=20, n_points=3000, pose_noise=(0.1, 0.01))
    last = len(w["poses"]) - 1
    win = list(range(last, last - W, -1))
    clouds = synthetic.ba_window_clouds(w, win, n_points=3000)
    want = oracle.local_ba_lidar(w["poses"], w["fixed"], w["points"], w["edges"], w["cam"], win, clouds, synthetic.TCL7, weight,
                                 iterations=10, lambda_init=lam)
    poses, pts, chi2, dpos, stats, ls = pkg.capi.local_lv_bundle_adjustment(w["poses"], w["fixed"], w["points"], pkg.pack_ba_edges(w["edges"]),
                                                                           w["cam"], win, clouds, synthetic.TCL7, weight, iterations=10,
                                                                           lambda_init=lam)
    assert ls.n_planes == want[6] and ls.n_planes > 10
    assert stats.iterations == want[4]
    assert stats.trials == int(want[5]["trials"].sum())
    assert abs(stats.final_chi2 - want[5]["chi2"][-1]) <= 1e-6 * want[5]["chi2"][-1]
    assert abs(ls.residual - want[7]["residual"]) <= 1e-6 * abs(want[7]["residual"])
    for k in range(len(poses)):
        assert rel_pose_err(poses[k], want[0][k]) < POSE_RTOL
    assert np.array_equal(poses[w["fixed"] > 0], w["poses"][w["fixed"] > 0])
    assert np.allclose(pts, want[1], rtol=POSE_RTOL, atol=1e-6)
    assert np.array_equal(dpos, want[3])
    assert np.allclose(chi2, want[2], rtol=1e-4, atol=1e-6)
    # the edge really takes part: the result differs from the visual-only optimisation
    base = pkg.local_bundle_adjustment(w["poses"], w["fixed"], w["points"], pkg.pack_ba_edges(w["edges"]), w["cam"], iterations=10, lambda_init=lam)
    assert np.abs(base[0] - poses).max() > 1e-8


def test_lidar_window_with_a_keyframe_outside_the_visual_graph(pkg, oracle, synthetic):
    """A window keyframe without any visual edge still becomes a variable (its vertex is connected through the LiDAR edge)."""
    w = synthetic.ba_window(5, n_opt=6, n_fix=6, n_points=800, pose_noise=(0.1, 0.01))
    last = len(w["poses"]) - 1
    win = [last, last - 1, last - 2]
    clouds = synthetic.ba_window_clouds(w, win, n_points=2400)
    edges = w["edges"][w["edges"][:, 1] != last - 1]
    seen = np.zeros(len(w["points"]), bool); seen[edges[:, 0].astype(int)] = True
    remap = np.cumsum(seen) - 1
    edges = edges.copy(); edges[:, 0] = remap[edges[:, 0].astype(int)]
    points = w["points"][seen]
    want = oracle.local_ba_lidar(w["poses"], w["fixed"], points, edges, w["cam"], win, clouds, synthetic.TCL7, 1.0)
    got = pkg.capi.local_lv_bundle_adjustment(w["poses"], w["fixed"], points, pkg.pack_ba_edges(edges), w["cam"], win, clouds, synthetic.TCL7, 1.0)
    assert got[4].iterations == want[4] and got[4].trials == int(want[5]["trials"].sum())
    for k in range(len(got[0])):
        assert rel_pose_err(got[0][k], want[0][k]) < POSE_RTOL
    assert np.abs(got[0][last - 1] - w["poses"][last - 1]).max() > 1e-9  # moved by the LiDAR edge alone


def test_lidar_window_argument_errors(pkg, synthetic):
    w = synthetic.ba_window(0, n_opt=3, n_fix=3, n_points=100)
    clouds = synthetic.ba_window_clouds(w, [5, 4], n_points=200)
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.lidar_window_evaluate(w["poses"], [5, 77], clouds, synthetic.TCL7)
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.lidar_window_evaluate(w["poses"], list(range(6)) * 4, clouds * 12, synthetic.TCL7)  # 24 keyframes > 20


def test_local_bundle_adjustment_batch(pkg, synthetic):
    """Windows optimised concurrently (own stream + workspace per host thread) give exactly the single-call results."""
    windows, singles = [], []
    for seed in range(7):
        w = synthetic.ba_window(seed, n_opt=6 + seed, n_fix=8, n_points=600 + 200 * seed, pose_noise=(0.1, 0.01))
        e = pkg.pack_ba_edges(w["edges"])
        d = dict(poses=w["poses"], fixed=w["fixed"], points=w["points"], edges=e)
        if seed % 2 == 0:
            last = len(w["poses"]) - 1
            win = list(range(last, last - 4, -1))
            d.update(win_pose=win, clouds=synthetic.ba_window_clouds(w, win, n_points=2000), Tcl7=synthetic.TCL7, weight=1.0)
            singles.append(pkg.capi.local_lv_bundle_adjustment(w["poses"], w["fixed"], w["points"], e, w["cam"], win, d["clouds"], synthetic.TCL7, 1.0))
        else:
            singles.append(pkg.local_bundle_adjustment(w["poses"], w["fixed"], w["points"], e, w["cam"]))
        windows.append(d)
        cam = w["cam"]
    # windows that stop early, start from a user lambda, or need several trials per iteration (large initial error)
    for seed, kw, noise in [(11, dict(iterations=3), (0.1, 0.01)), (12, dict(lambda_init=100.0), (0.1, 0.01)), (21, dict(), (3.0, 0.6)), (22, dict(), (1.0, 0.2)),
                            (14, dict(iterations=0), (0.1, 0.01))]:
        w = synthetic.ba_window(seed, n_opt=5, n_fix=6, n_points=500, pose_noise=noise)
        e = pkg.pack_ba_edges(w["edges"])
        d = dict(poses=w["poses"], fixed=w["fixed"], points=w["points"], edges=e, **kw)
        if seed in (21, 22):  # a heavy LiDAR edge: the optimiser rejects steps (the edge's b lacks the residual)
            last = len(w["poses"]) - 1
            win = list(range(last, last - 5, -1))
            d.update(win_pose=win, clouds=synthetic.ba_window_clouds(w, win, n_points=1500), Tcl7=synthetic.TCL7, weight=1000.0)
            singles.append(pkg.capi.local_lv_bundle_adjustment(w["poses"], w["fixed"], w["points"], e, w["cam"], win, d["clouds"], synthetic.TCL7, 1000.0))
        else:
            singles.append(pkg.local_bundle_adjustment(w["poses"], w["fixed"], w["points"], e, w["cam"], iterations=kw.get("iterations", 10),
                                                       lambda_init=kw.get("lambda_init", 0.0)))
        windows.append(d)
    assert max(s[4].trials - s[4].iterations for s in singles) > 3  # some windows had rejected steps
    batch = pkg.capi.BaBatch(windows, cam)
    for conc in (1, 4, 8):
        assert batch.run(max_concurrency=conc) == len(windows)
        for i, s in enumerate(singles):
            r = batch.result(i)
            assert batch.results[i] == s[4].iterations
            assert r[4].trials == s[4].trials, (conc, i)
            assert np.array_equal(r[0], s[0]) and np.array_equal(r[1], s[1]) and np.array_equal(r[3], s[3]), (conc, i)
            if s[4].iterations > 0:  # the per-edge chi2 is whatever the last error evaluation left; none ran with iterations = 0
                assert np.array_equal(r[2], s[2]), (conc, i)
            if len(s) > 5:
                assert r[5].n_planes == s[5].n_planes and r[5].residual == s[5].residual and r[5].hessian_evaluations == s[5].hessian_evaluations
            assert r[4].final_chi2 == s[4].final_chi2 and r[4].final_lambda == s[4].final_lambda and r[4].initial_chi2 == s[4].initial_chi2, (conc, i)
    # Round 6: the runs above took the Levenberg-Marquardt decisions ON THE DEVICE (k_ba_lm_begin_b / k_ba_lm_decide_b, rounds queued ahead
    # of the device; the singles run g2o's loop on the host).  The same batch with the decisions on the host between the phases
    # (TC2LI_BA_DEVICE_LM=0: rounds 2-5), with the host's LDL^T and with the device's (k_ba_solve_b repeats the host's operations in the host's
    # order): nothing changes -- trials, poses, points, per-edge chi2, the LiDAR edge's residual, bit for bit.
    import os
    for env in ({"TC2LI_BA_DEVICE_LM": "0"}, {"TC2LI_BA_DEVICE_LM": "0", "TC2LI_BA_DEVICE_SOLVE": "1"}):
        os.environ.update(env)
        try:
            assert batch.run(max_concurrency=8) == len(windows)
        finally:
            for k in env:
                del os.environ[k]
        for i, s in enumerate(singles):
            r = batch.result(i)
            assert batch.results[i] == s[4].iterations and r[4].trials == s[4].trials, (env, i)
            assert np.array_equal(r[0], s[0]) and np.array_equal(r[1], s[1]) and np.array_equal(r[3], s[3]), (env, i)
            if s[4].iterations > 0:
                assert np.array_equal(r[2], s[2]), (env, i)
            if len(s) > 5:
                assert r[5].n_planes == s[5].n_planes and r[5].residual == s[5].residual and r[5].hessian_evaluations == s[5].hessian_evaluations, (env, i)


def test_batch_with_a_large_window(pkg, synthetic):
    """More than 21 free poses (9 column tiles of the reduced system): the batch takes the per-tile Schur product; still identical
    to the single calls."""
    windows, singles = [], []
    for seed, n_opt in [(31, 24), (32, 6), (33, 14)]:
        w = synthetic.ba_window(seed, n_opt=n_opt, n_fix=5, n_points=900, pose_noise=(0.1, 0.01))
        e = pkg.pack_ba_edges(w["edges"])
        windows.append(dict(poses=w["poses"], fixed=w["fixed"], points=w["points"], edges=e))
        singles.append(pkg.local_bundle_adjustment(w["poses"], w["fixed"], w["points"], e, w["cam"]))
        cam = w["cam"]
    batch = pkg.capi.BaBatch(windows, cam)
    assert batch.run(max_concurrency=8) == len(windows)
    for i, s in enumerate(singles):
        r = batch.result(i)
        assert r[4].trials == s[4].trials and batch.results[i] == s[4].iterations
        assert np.array_equal(r[0], s[0]) and np.array_equal(r[1], s[1]) and np.array_equal(r[2], s[2])


@pytest.mark.parametrize("device_lm", ["1", "0"])  # the LM decisions on the device (rounds queued ahead) / on the host between the phases
def test_mixed_batch_falls_back_per_group(pkg, synthetic, device_lm, monkeypatch):
    """A batch of several lock-step groups in which ONE window lies outside the batched LiDAR kernels' range (8 keyframes in its LiDAR
    window, more than the 7 the lock-step kernels take): only that window's group goes through the one-window path -- every window of
    the batch, in the declined group and in the others, equals its one-window call bit for bit (the groups that had
    succeeded were optimised a second time, from their optimised state)."""
    monkeypatch.setenv("TC2LI_BA_DEVICE_LM", device_lm)
    windows, singles = [], []
    for seed in range(9):
        w = synthetic.ba_window(50 + seed, n_opt=8, n_fix=6, n_points=500, pose_noise=(0.1, 0.01))
        e = pkg.pack_ba_edges(w["edges"])
        last = len(w["poses"]) - 1
        win = list(range(last, last - (8 if seed == 4 else 4), -1))
        clouds = synthetic.ba_window_clouds(w, win, n_points=1200)
        windows.append(dict(poses=w["poses"], fixed=w["fixed"], points=w["points"], edges=e, win_pose=win, clouds=clouds, Tcl7=synthetic.TCL7, weight=1.0))
        singles.append(pkg.capi.local_lv_bundle_adjustment(w["poses"], w["fixed"], w["points"], e, w["cam"], win, clouds, synthetic.TCL7, 1.0))
        cam = w["cam"]
    assert all(s[4].iterations > 0 and s[4].final_chi2 < s[4].initial_chi2 for s in singles)
    batch = pkg.capi.BaBatch(windows, cam)
    for _ in range(2):  # the second call starts from the same inputs: BaBatch.run resets them
        assert batch.run(max_concurrency=8) == len(windows)
        for i, s in enumerate(singles):
            r = batch.result(i)
            assert batch.results[i] == s[4].iterations and r[4].trials == s[4].trials, i
            assert r[4].initial_chi2 == s[4].initial_chi2 and r[4].final_chi2 == s[4].final_chi2, i
            assert np.array_equal(r[0], s[0]) and np.array_equal(r[1], s[1]) and np.array_equal(r[2], s[2]) and np.array_equal(r[3], s[3]), i
            assert r[5].n_planes == s[5].n_planes and r[5].residual == s[5].residual, i


def test_lock_step_batch_against_the_oracle_at_the_benched_shape(pkg, oracle, synthetic):
    """The windows bench.py times (12 free + 20 fixed keyframes, 3000 points, LiDAR edge over 6 keyframes x 3000 points), through the
    lock-step batch entry, against the oracle directly: same iterations and LM trials, same planes, poses <= 1e-4 relative."""
    windows, wants = [], []
    for seed in (40, 41, 42, 43):
        w = synthetic.ba_window(seed, n_opt=12, n_fix=20, n_points=3000, pose_noise=(0.1, 0.01))
        last = len(w["poses"]) - 1
        win = list(range(last, last - 6, -1))
        clouds = synthetic.ba_window_clouds(w, win, n_points=3000)
        wants.append((w, oracle.local_ba_lidar(w["poses"], w["fixed"], w["points"], w["edges"], w["cam"], win, clouds, synthetic.TCL7, 1.0, iterations=10,
                                               lambda_init=0.0)))
        windows.append(dict(poses=w["poses"], fixed=w["fixed"], points=w["points"], edges=pkg.pack_ba_edges(w["edges"]), win_pose=win, clouds=clouds,
                            Tcl7=synthetic.TCL7, weight=1.0))
        cam = w["cam"]
    batch = pkg.capi.BaBatch(windows, cam)
    assert batch.run(max_concurrency=8) == len(windows)
    for i, (w, want) in enumerate(wants):
        poses, pts, chi2, dpos, stats, ls = batch.result(i)
        assert batch.results[i] == want[4] and stats.trials == int(want[5]["trials"].sum()), i
        assert ls.n_planes == want[6] and abs(ls.residual - want[7]["residual"]) <= 1e-6 * abs(want[7]["residual"])
        assert abs(stats.final_chi2 - want[5]["chi2"][-1]) <= 1e-6 * want[5]["chi2"][-1]
        for k in range(len(poses)):
            assert rel_pose_err(poses[k], want[0][k]) < POSE_RTOL, (i, k)
        assert np.allclose(pts, want[1], rtol=POSE_RTOL, atol=1e-6) and np.array_equal(dpos, want[3])


def varied_window_dict(pkg, synthetic, w):
    d = dict(poses=w["poses"], fixed=w["fixed"], points=w["points"], edges=pkg.pack_ba_edges(w["edges"]), iterations=w["iterations"])
    if w["win_pose"]:
        d.update(win_pose=w["win_pose"], clouds=w["clouds"], Tcl7=synthetic.TCL7, weight=w["weight"])
    return d


def test_mixed_lock_step_batch_against_the_oracle_window_by_window(pkg, oracle, synthetic):
    """a 16-window batch from the generator bench.py's headline uses (synthetic.ba_window_varied: 4-24 free and 2-40 fixed
    keyframes, 500-6000 points, 0-15 % outliers, LiDAR windows of 0 / 3-6 clouds, heavy LiDAR edges whose steps are rejected, windows
    interrupted after 2-7 iterations), through ONE lock-step group -- the Levenberg-Marquardt loop on the device, every window at its
    own pace -- against the oracle window by window: the same iterations and trials, the same planes, poses <= 1e-4 relative.  The batch holds
    windows of the sparse and of the dense Schur path (more than 21 free keyframes), windows that end early and windows that retry."""
    seeds = list(range(14)) + [17, 39]   # (the two heavy LiDAR edges of the benched 64 among them)
    ws = [synthetic.ba_window_varied(s) for s in seeds]
    assert any(w["params"]["n_opt"] > 21 for w in ws) and any(not w["win_pose"] for w in ws) and any(w["weight"] > 1 for w in ws)
    batch = pkg.capi.BaBatch([varied_window_dict(pkg, synthetic, w) for w in ws], ws[0]["cam"])
    assert batch.run_group(0) == len(ws)
    its, extra_trials = [], 0
    for i, w in enumerate(ws):
        if w["win_pose"]:
            want = oracle.local_ba_lidar(w["poses"], w["fixed"], w["points"], w["edges"], w["cam"], w["win_pose"], w["clouds"], synthetic.TCL7, w["weight"],
                                         iterations=w["iterations"])
        else:
            want = oracle.local_ba(w["poses"], w["fixed"], w["points"], w["edges"], w["cam"], iterations=w["iterations"])
        r = batch.result(i)
        poses, pts, chi2, dpos, stats = r[:5]
        assert batch.results[i] == want[4] and stats.trials == int(want[5]["trials"].sum()), (i, w["params"], batch.results[i], want[4], stats.trials, want[5]["trials"])
        if w["win_pose"]:
            assert r[5].n_planes == want[6] and abs(r[5].residual - want[7]["residual"]) <= 1e-6 * max(abs(want[7]["residual"]), 1e-9), i
        if want[4] > 0:
            assert abs(stats.final_chi2 - want[5]["chi2"][-1]) <= 1e-6 * want[5]["chi2"][-1], i
        for k in range(len(poses)):
            assert rel_pose_err(poses[k], want[0][k]) < POSE_RTOL, (i, k)
        assert np.allclose(pts, want[1], rtol=POSE_RTOL, atol=1e-6) and np.array_equal(dpos, want[3]), i
        its.append(want[4])
        extra_trials += int(want[5]["trials"].sum()) - want[4]
    assert min(its) < 10 and max(its) == 10 and extra_trials > 0, (its, extra_trials)  # windows that stop early, windows with rejected steps


def _window_results(batch, i):
    r = batch.result(i)
    return tuple(np.array(a, copy=True) for a in r[:4]) + (int(batch.results[i]), r[4].iterations, r[4].trials, r[4].initial_chi2, r[4].final_chi2,
                                                           r[5].n_planes, r[5].residual, r[5].chi2)


@pytest.mark.parametrize("knobs", [{}, {"TC2LI_BA_ENGINE_SIDE": "1", "TC2LI_BA_ENGINE_STAGE": "1,0"}, {"TC2LI_BA_ENGINE_STAGE": "3,5"}])
def test_engine_gives_every_window_the_bits_of_the_batch_call(pkg, synthetic, monkeypatch, knobs):
    """tc2li_ba_engine (windows join and leave one running lock-step queue): the varied mix -- camera-only and LiDAR windows, sparse and dense
    Schur forms, windows that end early -- through an engine with FEWER slots than windows, as two tickets of which the second is
    submitted while the first runs; every window's poses, points, per-edge chi2, depth flags and statistics are bit for bit the ones of
    tc2li_local_bundle_adjustment_batch_group.  An empty ticket returns at once; a ticket is collected once."""
    for k, v in knobs.items():   # (an engine reads its switches when it is created: the plane extraction on a side stream, how windows are staged)
        monkeypatch.setenv(k, v)
    seeds = list(range(14)) + [17, 39]
    ws = [synthetic.ba_window_varied(s) for s in seeds]
    dicts = [varied_window_dict(pkg, synthetic, w) for w in ws]
    ref = pkg.capi.BaBatch(dicts, ws[0]["cam"])
    assert ref.run_group(0) == len(ws)
    want = [_window_results(ref, i) for i in range(len(ws))]
    for cap in (5, 64):
        eng = pkg.capi.BaEngine(ws[0]["cam"], max_windows=cap)
        b = pkg.capi.BaBatch(dicts, ws[0]["cam"])
        t0 = eng.submit(b, 0, 9)
        t1 = eng.submit(b, 9, len(ws) - 9)
        te = eng.submit(b, 0, 0)
        assert eng.wait(te) == 0
        assert eng.wait(t1) == len(ws) - 9 and eng.wait(t0) == 9
        with pytest.raises(pkg.capi.Tc2liError):
            eng.wait(t0)
        for i in range(len(ws)):
            got = _window_results(b, i)
            for a, c in zip(got, want[i]):
                assert np.array_equal(a, c), (cap, i, ws[i]["params"])
        # the engine is reusable: the same windows again
        t2 = eng.submit(b)
        assert eng.wait(t2) == len(ws)
        for i in range(len(ws)):
            for a, c in zip(_window_results(b, i), want[i]):
                assert np.array_equal(a, c), (cap, i)
        eng.close()


def test_engine_edge_cases(pkg, synthetic):
    """The engine beside the one-window call on the windows the batched kernels do not take and on the calls' error behaviour: a window of 27
    free keyframes (outside the device-side LM: the one-window path inside the engine), a window whose stop flag is set (nothing moves, 0
    iterations: OptimizerWithLidar.cc:387-391), a window with every pose fixed, a malformed window (its result is the error code, its
    neighbours are optimised), tickets from several threads at once."""
    import threading
    cam = synthetic.ba_window(0, n_opt=4, n_fix=2, n_points=100)["cam"]

    def as_dict(w, **kw):
        return dict(poses=w["poses"], fixed=w["fixed"], points=w["points"], edges=pkg.pack_ba_edges(w["edges"]), **kw)
    wide = synthetic.ba_window(9, n_opt=27, n_fix=5, n_points=1500)
    small = synthetic.ba_window(3, n_opt=5, n_fix=4, n_points=300)
    allfix = dict(small, fixed=np.ones_like(small["fixed"]))
    bad = as_dict(small)
    bad["edges"] = bad["edges"].copy(); bad["edges"]["pose"][0] = 10 ** 6
    dicts = [as_dict(wide), as_dict(small, stop_flag=np.ones(1, np.uint8)), as_dict(allfix, iterations=5), bad, as_dict(small)]
    eng = pkg.capi.BaEngine(cam, max_windows=3)
    b = pkg.capi.BaBatch(dicts, cam)
    assert eng.wait(eng.submit(b)) == 4
    assert b.results[3] < 0 and b.results[1] == 0 and b.result(1)[4].iterations == 0
    assert np.array_equal(b.result(1)[0], small["poses"]) and np.array_equal(b.result(1)[1], small["points"])
    for i, w, kw in [(0, wide, {}), (2, allfix, dict(iterations=5)), (4, small, {})]:
        poses, pts, chi2, dpos, stats = pkg.local_bundle_adjustment(w["poses"], w["fixed"], w["points"], pkg.pack_ba_edges(w["edges"]), cam, **kw)
        got = b.result(i)
        assert b.results[i] == stats.iterations and got[4].trials == stats.trials and got[4].n_free_poses == stats.n_free_poses, i
        if i == 0:   # (the one-window path's host-side LM and the device-side LM agree to the tolerance, not bit for bit)
            assert np.array_equal(got[0], poses) and np.array_equal(got[1], pts) and np.array_equal(got[2], chi2) and np.array_equal(got[3], dpos)
        else:
            assert np.allclose(got[0], poses, rtol=1e-9, atol=1e-12) and np.allclose(got[1], pts, rtol=1e-9, atol=1e-12) and np.array_equal(got[3], dpos), i
    # several submitters: every thread's windows come back with the bits of the first run
    want = [np.array(b.result(4)[0], copy=True), np.array(b.result(4)[1], copy=True)]
    batches = [pkg.capi.BaBatch([as_dict(small)] * 4, cam) for _ in range(4)]
    out = {}

    def submitter(k):
        out[k] = [eng.wait(eng.submit(batches[k])) for _ in range(3)]
    ts = [threading.Thread(target=submitter, args=(k,)) for k in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert out == {k: [4, 4, 4] for k in range(4)}
    for bb in batches:
        for i in range(4):
            assert np.array_equal(bb.result(i)[0], want[0]) and np.array_equal(bb.result(i)[1], want[1])
    # destroying an engine with work in its queue finishes that work first
    t = eng.submit(batches[0])
    eng.close()
    assert all(batches[0].results[i] == b.results[4] for i in range(4)) and t > 0
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.BaEngine(cam, max_windows=0)


def test_group_call_equals_the_batch_call(pkg, synthetic):
    """tc2li_local_bundle_adjustment_batch_group (one lock-step group on a context of the caller's choice: the mapping workers of a
    multi-sequence system) gives every window the result of the common batch call, bit for bit; two groups side by side do not disturb
    each other; a group number outside 0 .. 7 is refused."""
    import threading
    windows = []
    for seed in range(6):
        w = synthetic.ba_window(30 + seed, n_opt=8, n_fix=8, n_points=700, pose_noise=(0.1, 0.01))
        d = dict(poses=w["poses"], fixed=w["fixed"], points=w["points"], edges=pkg.pack_ba_edges(w["edges"]))
        if seed % 2 == 0:
            last = len(w["poses"]) - 1
            win = list(range(last, last - 4, -1))
            d.update(win_pose=win, clouds=synthetic.ba_window_clouds(w, win, n_points=2000), Tcl7=synthetic.TCL7, weight=1.0)
        windows.append(d)
        cam = w["cam"]
    ref = pkg.capi.BaBatch(windows, cam)
    assert ref.run(max_concurrency=8) == len(windows)
    want = [tuple(np.array(a, copy=True) for a in ref.result(i)[:4]) + (ref.result(i)[4].iterations, ref.result(i)[4].trials, ref.result(i)[4].final_chi2)
            for i in range(len(windows))]
    a, b = pkg.capi.BaBatch(windows, cam), pkg.capi.BaBatch(windows, cam)
    got = {}
    ts = [threading.Thread(target=lambda bb=bb, g=g: got.__setitem__(g, bb.run_group(g))) for bb, g in ((a, 1), (b, 5))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert got == {1: len(windows), 5: len(windows)}
    for batch in (a, b):
        for i, wnt in enumerate(want):
            r = batch.result(i)
            assert all(np.array_equal(r[k], wnt[k]) for k in range(4)), i
            assert (r[4].iterations, r[4].trials, r[4].final_chi2) == wnt[4:], i
    with pytest.raises(pkg.Tc2liError):
        a.run_group(8)
