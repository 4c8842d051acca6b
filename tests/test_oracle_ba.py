"""CPU tests of the optimisation oracle: analytic Jacobians against g2o's numeric recipe (central differences,
delta 1e-9, base_binary_edge.hpp:164-214), one Levenberg step with Schur complement against a dense numpy solve of
the full system, convergence towards the synthetic ground truth, PoseOptimization outlier handling."""
import numpy as np
import pytest

S5991, S7815 = np.float32(np.sqrt(5.991)), np.float32(np.sqrt(7.815))


def huber(chi2, delta):
    dsqr = np.float32(np.float64(delta) * np.float64(delta))
    if chi2 <= dsqr:
        return chi2, 1.0
    s = np.sqrt(chi2)
    return 2 * s * float(delta) - dsqr, float(delta) / s


@pytest.fixture(scope="module")
def small(synthetic):
    return synthetic.ba_window(3, n_opt=4, n_fix=2, n_points=80)


def test_jacobians_against_numeric(oracle, small):
    w = small
    rng = np.random.default_rng(0)
    for e in w["edges"][rng.choice(len(w["edges"]), 40, replace=False)]:
        pose, X = w["poses"][int(e[1])], w["points"][int(e[0])]
        err, A, B = oracle.edge_linearize(pose, X, e, w["cam"])
        d = 1e-3  # large enough to average out the float-precision invz of the stereo edge (types_six_dof_expmap.cpp:191)
        for c in range(3):
            dx = np.zeros(3); dx[c] = d
            e1 = oracle.edge_linearize(pose, X + dx, e, w["cam"])[0]
            e0 = oracle.edge_linearize(pose, X - dx, e, w["cam"])[0]
            assert np.allclose((e1 - e0) / (2 * d), A[:, c], rtol=1e-3, atol=3e-2)
        for c in range(6):
            u = np.zeros(6); u[c] = d
            e1 = oracle.edge_linearize(oracle.se3_exp_mul(u, pose), X, e, w["cam"])[0]
            e0 = oracle.edge_linearize(oracle.se3_exp_mul(-u, pose), X, e, w["cam"])[0]
            assert np.allclose((e1 - e0) / (2 * d), B[:, c], rtol=1e-3, atol=3e-2)
        assert len(err) == (3 if e[4] >= 0 else 2)


def test_one_lm_step_matches_dense_solve(oracle, small):
    w = small
    K, P = len(w["poses"]), len(w["points"])
    var = np.nonzero(w["fixed"] == 0)[0]
    pidx = {k: i for i, k in enumerate(var)}
    n = 6 * len(var) + 3 * P
    H, b = np.zeros((n, n)), np.zeros(n)
    for e in w["edges"]:
        p, k = int(e[0]), int(e[1])
        err, A, B = oracle.edge_linearize(w["poses"][k], w["points"][p], e, w["cam"])
        chi2 = e[5] * err @ err
        _, r1 = huber(chi2, S7815 if e[4] >= 0 else S5991)
        wgt = r1 * e[5]
        lp = 6 * len(var) + 3 * p
        H[lp:lp + 3, lp:lp + 3] += wgt * A.T @ A
        b[lp:lp + 3] += -wgt * A.T @ err
        if k in pidx:
            kp = 6 * pidx[k]
            H[kp:kp + 6, kp:kp + 6] += wgt * B.T @ B
            H[kp:kp + 6, lp:lp + 3] += wgt * B.T @ A
            H[lp:lp + 3, kp:kp + 6] += wgt * A.T @ B
            b[kp:kp + 6] += -wgt * B.T @ err
    lam = 1e-5 * np.abs(np.diag(H)).max()
    x = np.linalg.solve(H + lam * np.eye(n), b)
    poses, pts, _, _, it, tr = oracle.local_ba(w["poses"], w["fixed"], w["points"], w["edges"], w["cam"], iterations=1)
    assert it == 1 and tr["trials"][0] == 1  # first trial accepted
    for k in var:
        want = oracle.se3_exp_mul(x[6 * pidx[k]:6 * pidx[k] + 6], w["poses"][k])
        assert np.allclose(poses[k], want, rtol=0, atol=1e-9)
    assert np.allclose(pts, w["points"] + x[6 * len(var):].reshape(-1, 3), rtol=0, atol=1e-8)
    assert np.array_equal(poses[w["fixed"] > 0], w["poses"][w["fixed"] > 0])


def test_ba_converges_towards_truth(oracle, synthetic):
    w = synthetic.ba_window(1, n_opt=8, n_fix=10, n_points=800, outlier_frac=0.0, mono_frac=0.1)
    poses, pts, chi2, dpos, it, tr = oracle.local_ba(w["poses"], w["fixed"], w["points"], w["edges"], w["cam"], iterations=10)
    assert np.all(np.diff(tr["chi2"]) <= 1e-9) and tr["chi2"][-1] < 0.9 * tr["chi2"][0]
    opt = w["fixed"] == 0
    e0 = np.abs(w["poses"][opt, 4:] - w["poses_true"][opt, 4:]).max()
    e1 = np.abs(poses[opt, 4:] - w["poses_true"][opt, 4:]).max()
    assert e1 < 0.5 * e0
    assert np.all(dpos == 1)
    # with lambda_init given (the inertial-map branch uses 100, OptimizerWithLidar.cc:141-142) the trace starts there
    tr2 = oracle.local_ba(w["poses"], w["fixed"], w["points"], w["edges"], w["cam"], iterations=3, lambda_init=100.0)[5]
    assert abs(tr2["lam"][0] - 100.0 / 3) < 1e-9 or tr2["lam"][0] >= 100.0 / 3


def test_pose_optimization(oracle, synthetic):
    w = synthetic.ba_window(2, n_opt=6, n_fix=6, n_points=1500, outlier_frac=0.08)
    k = len(w["poses"]) - 1
    ed = w["edges"][w["edges"][:, 1] == k].copy()
    Xw = w["points_true"][ed[:, 0].astype(int)]
    ed[:, 0] = np.arange(len(ed)); ed[:, 1] = 0
    pose, out, inl, tr = oracle.pose_optimization(w["poses"][k], Xw, ed, w["cam"])
    assert inl == len(ed) - out.sum() and 0.02 < out.mean() < 0.25
    assert np.abs(pose[4:] - w["poses_true"][k][4:]).max() < 0.3 * np.abs(w["poses"][k][4:] - w["poses_true"][k][4:]).max()
    assert np.array_equal(pose, pose.astype(np.float32).astype(np.float64))  # SetPose stores float
    # fewer than 3 correspondences: returns 0 and leaves the pose alone (Optimizer.cc:999-1000)
    p2, o2, inl2, _ = oracle.pose_optimization(w["poses"][k], Xw[:2], ed[:2], w["cam"])
    assert inl2 == 0 and np.array_equal(p2, w["poses"][k])
