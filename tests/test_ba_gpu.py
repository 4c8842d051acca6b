"""GPU parity of the optimisation back end with the oracle.  Bar (BASELINE.json): optimised SE3 poses within 1e-4
relative; here the inlier/outlier sets must also be identical and the poses agree to ~1e-9 (the only difference is
the order of the floating-point reductions)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POSE_RTOL = 1e-4  # the tolerance BASELINE.json states for optimised SE3 poses


def frame_problem(w, k, truth_points=True):
    ed = w["edges"][w["edges"][:, 1] == k].copy()
    Xw = (w["points_true"] if truth_points else w["points"])[ed[:, 0].astype(int)]
    ed[:, 0] = np.arange(len(ed)); ed[:, 1] = 0
    return Xw, ed


@pytest.mark.parametrize("seed,outliers", [(2, 0.08), (5, 0.0), (7, 0.3)])
def test_pose_optimization(pkg, oracle, synthetic, seed, outliers):
    w = synthetic.ba_window(seed, n_opt=6, n_fix=6, n_points=1500, outlier_frac=outliers)
    for k in (len(w["poses"]) - 1, len(w["poses"]) - 3):
        Xw, ed = frame_problem(w, k)
        want_pose, want_out, want_inl, _ = oracle.pose_optimization(w["poses"][k], Xw, ed, w["cam"])
        pose, out, inl = pkg.pose_optimization(w["poses"][k], Xw, pkg.pack_ba_edges(ed), w["cam"])
        assert inl == want_inl
        assert np.array_equal(out, want_out)
        assert np.allclose(pose, want_pose, rtol=POSE_RTOL, atol=1e-7)
        assert np.abs(pose - want_pose).max() < 1e-6  # in practice identical up to the float rounding of SetPose


def test_pose_optimization_edge_cases(pkg, oracle, synthetic):
    w = synthetic.ba_window(3, n_opt=3, n_fix=3, n_points=400)
    k = len(w["poses"]) - 1
    Xw, ed = frame_problem(w, k)
    for n in (0, 2, 3, 9, 10, 40):
        want_pose, want_out, want_inl, _ = oracle.pose_optimization(w["poses"][k], Xw[:n], ed[:n], w["cam"])
        pose, out, inl = pkg.pose_optimization(w["poses"][k], Xw[:n], pkg.pack_ba_edges(ed[:n]), w["cam"])
        assert inl == want_inl and np.array_equal(out, want_out)
        assert np.allclose(pose, want_pose, rtol=POSE_RTOL, atol=1e-7)
    # monocular-only correspondences
    edm = ed.copy(); edm[:, 4] = -1
    want_pose, want_out, want_inl, _ = oracle.pose_optimization(w["poses"][k], Xw, edm, w["cam"])
    pose, out, inl = pkg.pose_optimization(w["poses"][k], Xw, pkg.pack_ba_edges(edm), w["cam"])
    assert inl == want_inl and np.array_equal(out, want_out) and np.allclose(pose, want_pose, rtol=POSE_RTOL, atol=1e-7)


def test_pose_optimization_batch(pkg, oracle, synthetic):
    w = synthetic.ba_window(4, n_opt=8, n_fix=4, n_points=1200, outlier_frac=0.1)
    ks = list(range(4, 12))
    probs = [frame_problem(w, k) for k in ks]
    offs = np.concatenate([[0], np.cumsum([len(e) for _, e in probs])])
    Xw = np.concatenate([x for x, _ in probs])
    ed = np.concatenate([e for _, e in probs])
    poses, out, inl = pkg.pose_optimization_batch(w["poses"][ks], offs, Xw, pkg.pack_ba_edges(ed), w["cam"])
    for i, k in enumerate(ks):
        want_pose, want_out, want_inl, _ = oracle.pose_optimization(w["poses"][k], probs[i][0], probs[i][1], w["cam"])
        assert inl[i] == want_inl
        assert np.array_equal(out[offs[i]:offs[i + 1]], want_out)
        assert np.allclose(poses[i], want_pose, rtol=POSE_RTOL, atol=1e-7)


def rel_pose_err(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


@pytest.mark.parametrize("seed,n_opt,n_fix,n_pts,lam", [(0, 12, 20, 3000, 0.0), (1, 4, 2, 120, 0.0), (2, 8, 10, 800, 100.0),
                                                          (6, 20, 12, 2000, 0.0), (7, 24, 6, 1500, 0.0),   # 24: the lean form with two workgroups per part
                                                          (9, 27, 5, 1500, 0.0),                          # 27: the block-sparse MFMA kernels
                                                          (8, 15, 6, 1200, 0.0)])  # 15 free keyframes: (2, 2) ranges, the lean form's largest closing area
def test_local_bundle_adjustment(pkg, oracle, synthetic, seed, n_opt, n_fix, n_pts, lam):
    w = synthetic.ba_window(seed, n_opt=n_opt, n_fix=n_fix, n_points=n_pts)
    want = oracle.local_ba(w["poses"], w["fixed"], w["points"], w["edges"], w["cam"], iterations=10, lambda_init=lam)
    poses, pts, chi2, dpos, stats = pkg.local_bundle_adjustment(w["poses"], w["fixed"], w["points"], pkg.pack_ba_edges(w["edges"]),
                                                              w["cam"], iterations=10, lambda_init=lam)
    assert stats.iterations == want[4]
    assert stats.trials == int(want[5]["trials"].sum())
    assert abs(stats.final_chi2 - want[5]["chi2"][-1]) <= 1e-6 * want[5]["chi2"][-1]
    # optimised SE3 poses within 1e-4 relative (BASELINE.json); points likewise
    for k in range(len(poses)):
        assert rel_pose_err(poses[k], want[0][k]) < POSE_RTOL
    assert np.array_equal(poses[w["fixed"] > 0], w["poses"][w["fixed"] > 0])
    assert np.allclose(pts, want[1], rtol=POSE_RTOL, atol=1e-6)
    # same outlier set under the reference's rules (OptimizerWithLidar.cc:406-449)
    stereo = w["edges"][:, 4] >= 0
    th = np.where(stereo, 7.815, 5.991)
    assert np.array_equal((chi2 > th) | (dpos == 0), (want[2] > th) | (want[3] == 0))
    assert np.allclose(chi2, want[2], rtol=1e-5, atol=1e-7)


def test_local_ba_stop_flag_and_structure(pkg, oracle, synthetic):
    w = synthetic.ba_window(3, n_opt=5, n_fix=4, n_points=300)
    e = pkg.pack_ba_edges(w["edges"])
    # *pbStopFlag set before the optimisation: nothing moves (OptimizerWithLidar.cc:387-391)
    stop = np.ones(1, np.uint8)
    poses, pts, chi2, dpos, stats = pkg.local_bundle_adjustment(w["poses"], w["fixed"], w["points"], e, w["cam"], stop_flag=stop)
    assert stats.iterations == 0 and np.array_equal(poses, w["poses"]) and np.array_equal(pts, w["points"])
    # every pose fixed: only the landmarks move
    allfix = np.ones_like(w["fixed"])
    want = oracle.local_ba(w["poses"], allfix, w["points"], w["edges"], w["cam"], iterations=5)
    poses, pts, chi2, dpos, stats = pkg.local_bundle_adjustment(w["poses"], allfix, w["points"], e, w["cam"], iterations=5)
    assert np.array_equal(poses, w["poses"]) and stats.n_free_poses == 0
    assert np.allclose(pts, want[1], rtol=POSE_RTOL, atol=1e-6)
    # malformed input is refused
    bad = e.copy(); bad["pose"][0] = 10 ** 6
    with pytest.raises(pkg.Tc2liError):
        pkg.local_bundle_adjustment(w["poses"], w["fixed"], w["points"], bad, w["cam"])
    # two (and three) edges between the same point and the same free keyframe: g2o adds their blocks to the same Hpl / Hpp entries
    # (base_binary_edge.hpp:55-137) -- so does the library since round 5 (k_ba_dups: the later edges' blocks on top of the first one's slot;
    # rounds 2-4 refused such a window): the oracle's result, alone and in a lock-step batch beside a window without duplicates
    free = np.flatnonzero(w["fixed"] == 0)
    ks = [int(np.flatnonzero(e["pose"] == int(free[0]))[0]), int(np.flatnonzero(e["pose"] == int(free[0]))[0]),
          int(np.flatnonzero(e["pose"] == int(free[1]))[3]), int(np.flatnonzero(e["pose"] == int(free[-1]))[7])]
    w6 = np.concatenate([w["edges"]] + [w["edges"][k:k + 1] for k in ks])
    w6[-1, 2] += 0.75   # the repeated observations need not agree with the first ones
    w6[-2, 3] -= 0.5
    dup = pkg.pack_ba_edges(w6)
    want = oracle.local_ba(w["poses"], w["fixed"], w["points"], w6, w["cam"], iterations=6)
    poses, pts, chi2, dpos, stats = pkg.local_bundle_adjustment(w["poses"], w["fixed"], w["points"], dup, w["cam"], iterations=6)
    assert stats.iterations == want[4] and stats.trials == int(want[5]["trials"].sum())
    assert abs(stats.final_chi2 - want[5]["chi2"][-1]) <= 1e-6 * want[5]["chi2"][-1]
    assert np.allclose(poses, want[0], rtol=POSE_RTOL, atol=1e-7) and np.allclose(pts, want[1], rtol=POSE_RTOL, atol=1e-6)
    assert np.allclose(chi2, want[2], rtol=1e-5, atol=1e-7)
    batch = pkg.capi.BaBatch([dict(poses=w["poses"], fixed=w["fixed"], points=w["points"], edges=dup, iterations=6),
                              dict(poses=w["poses"], fixed=w["fixed"], points=w["points"], edges=e, iterations=6),
                              dict(poses=w["poses"], fixed=w["fixed"], points=w["points"], edges=dup, iterations=6)], w["cam"])
    assert batch.run(8) == 3
    for i in (0, 2):
        assert np.array_equal(batch.result(i)[0], poses) and np.array_equal(batch.result(i)[1], pts)
    plain = pkg.local_bundle_adjustment(w["poses"], w["fixed"], w["points"], e, w["cam"], iterations=6)
    assert np.array_equal(batch.result(1)[0], plain[0]) and np.array_equal(batch.result(1)[1], plain[1])
    # the same pair on a FIXED keyframe only adds to the landmark's block: accepted
    fixed_pose = int(np.flatnonzero(w["fixed"] > 0)[0])
    k = int(np.flatnonzero(e["pose"] == fixed_pose)[0])
    dup = np.concatenate([e, e[k:k + 1]])
    w6 = np.concatenate([w["edges"], w["edges"][k:k + 1]])
    want = oracle.local_ba(w["poses"], w["fixed"], w["points"], w6, w["cam"], iterations=5)
    poses, pts, chi2, dpos, stats = pkg.local_bundle_adjustment(w["poses"], w["fixed"], w["points"], dup, w["cam"], iterations=5)
    assert np.allclose(poses, want[0], rtol=POSE_RTOL, atol=1e-7)
