"""CPU tests of the visual-inertial BA oracle: analytic Jacobians of the inertial edge and
of the body-frame projection edges against finite differences with the reference's own update rules (ImuCamPose::Update,
additive velocity / bias vertices), zero residual on a consistent trajectory, behaviour of LocalInertialBA on a synthetic window."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def window(synthetic, oracle):
    w = synthetic.inertial_window(0, n_opt=6, n_points=400)
    pre = []
    for s, t1, t2 in w["samples"]:
        _, f = oracle.imu_preintegrate(s, t1, t2, w["bias6"], *synthetic.IMU_NOISE)
        pre.append(oracle.pack_preintegrated(f, w["bias6"]))
    w["pre298"] = np.stack(pre)
    return w


def test_inertial_residual_is_small_on_the_true_trajectory(oracle, window):
    w = window
    for l, (k1, k2) in enumerate(w["link4"][:, :2].astype(int)):
        err, _ = oracle.inertial_edge(w["kf33_true"][k1], w["kf33_true"][k2], w["pre298"][l])
        assert np.abs(err[:3]).max() < 2e-3 and np.abs(err[3:6]).max() < 2e-2 and np.abs(err[6:]).max() < 1e-2
        noisy, _ = oracle.inertial_edge(w["kf33"][k1], w["kf33"][k2], w["pre298"][l])
        assert np.abs(noisy).max() > 5 * np.abs(err).max()


def _perturb(oracle, kf, calib, col, d):
    """Apply the vertex update of column `col` of the 24-column layout (P1 6 | V1 3 | G1 3 | A1 3 | P2 6 | V2 3) to keyframe state(s)."""
    k1, k2 = kf[0].copy(), kf[1].copy()
    if col < 6:
        u = np.zeros(6); u[col] = d
        k1, _ = oracle.imu_pose_update(k1, 0, calib, u)
    elif col < 15:
        k1[24 + (col - 6)] += d
    elif col < 21:
        u = np.zeros(6); u[col - 15] = d
        k2, _ = oracle.imu_pose_update(k2, 0, calib, u)
    else:
        k2[24 + (col - 21)] += d
    return k1, k2


def test_inertial_edge_jacobians(oracle, window):
    w = window
    l = 2
    k1, k2 = w["link4"][l, :2].astype(int)
    kf = (w["kf33"][k1], w["kf33"][k2])
    err, J = oracle.inertial_edge(kf[0], kf[1], w["pre298"][l])
    for col in range(24):
        d = 1e-4 if 9 <= col < 15 else 1e-6   # the bias enters through float arithmetic (GetDeltaRotation etc.): larger step
        a = oracle.inertial_edge(*_perturb(oracle, kf, w["calib24"], col, +d), w["pre298"][l])[0]
        b = oracle.inertial_edge(*_perturb(oracle, kf, w["calib24"], col, -d), w["pre298"][l])[0]
        fd = (a - b) / (2 * d)
        tol = 2e-2 if 9 <= col < 15 else 1e-4
        assert np.allclose(fd, J[:, col], rtol=tol, atol=tol * max(1.0, np.abs(J[:, col]).max())), col


def test_visual_edge_jacobians(oracle, window):
    w = window
    rng = np.random.default_rng(1)
    for e in w["edges"][rng.choice(len(w["edges"]), 12, replace=False)]:
        kf = w["kf33"][int(e[1])]
        X = w["points"][int(e[0])]
        dim, err, A, B = oracle.inertial_visual_edge(kf, w["calib24"], X, e, w["cam"])
        assert dim == (3 if e[4] >= 0 else 2)
        d = 1e-6
        for c in range(3):
            dx = np.zeros(3); dx[c] = d
            fd = (oracle.inertial_visual_edge(kf, w["calib24"], X + dx, e, w["cam"])[1] - oracle.inertial_visual_edge(kf, w["calib24"], X - dx, e, w["cam"])[1]) / (2 * d)
            assert np.allclose(fd[:dim], A[:dim, c], rtol=1e-5, atol=1e-5)
        for c in range(6):
            u = np.zeros(6); u[c] = d
            kp, _ = oracle.imu_pose_update(kf, 0, w["calib24"], u)
            km, _ = oracle.imu_pose_update(kf, 0, w["calib24"], -u)
            fd = (oracle.inertial_visual_edge(kp, w["calib24"], X, e, w["cam"])[1] - oracle.inertial_visual_edge(km, w["calib24"], X, e, w["cam"])[1]) / (2 * d)
            assert np.allclose(fd[:dim], B[:dim, c], rtol=1e-4, atol=1e-4 * max(1.0, np.abs(B).max())), c


def test_pose_update_keeps_the_camera_consistent(oracle, window):
    w = window
    kf = w["kf33"][3]
    its = 0
    for step in range(7):
        kf, its = oracle.imu_pose_update(kf, its, w["calib24"], np.array([0.01, -0.02, 0.015, 0.05, 0.02, -0.01]))
        Rcw, tcw, Rwb, twb = kf[:9].reshape(3, 3), kf[9:12], kf[12:21].reshape(3, 3), kf[21:24]
        Rcb, tcb = w["calib24"][:9].reshape(3, 3), w["calib24"][9:12]
        assert np.allclose(Rcw, Rcb @ Rwb.T, atol=1e-12) and np.allclose(tcw, Rcb @ (-Rwb.T @ twb) + tcb, atol=1e-12)
        assert np.allclose(Rwb @ Rwb.T, np.eye(3), atol=1e-12 if its == 0 else 1e-6)  # the input is a float matrix; normalised every third update
        assert its == (step + 1) % 3  # NormalizeRotation every third update


def test_local_inertial_ba_reduces_the_error(oracle, window):
    w = window
    kf, pts, chi2, dpos, it, trace, (err, err_end) = oracle.local_inertial_ba(w["kf33"], w["fixed"], w["has_imu"], w["calib24"], w["points"], w["edges"],
                                                                             w["link4"], w["pre298"], w["cam"], iterations=10, lambda_init=1.0)
    assert it >= 3 and err_end < 0.2 * err
    assert np.all(np.diff(trace["chi2"]) <= 1e-9)
    def pos_err(a):
        return np.linalg.norm(a[:, 21:24] - w["kf33_true"][:, 21:24], axis=1).mean()
    assert pos_err(kf) < 0.5 * pos_err(w["kf33"])
    assert np.array_equal(kf[0], w["kf33"][0])           # the fixed keyframe (pose, velocity, biases)
    vel_err0 = np.linalg.norm(w["kf33"][1:, 24:27] - w["kf33_true"][1:, 24:27], axis=1).mean()
    vel_err1 = np.linalg.norm(kf[1:, 24:27] - w["kf33_true"][1:, 24:27], axis=1).mean()
    assert vel_err1 < vel_err0
    inl = chi2 < 7.815
    assert inl.mean() > 0.9 and dpos.all()
    # the large-window settings of the reference (4 iterations, lambda 1e-2) also run
    r2 = oracle.local_inertial_ba(w["kf33"], w["fixed"], w["has_imu"], w["calib24"], w["points"], w["edges"], w["link4"], w["pre298"], w["cam"], 4, 1e-2)
    assert r2[4] >= 1 and r2[6][1] < r2[6][0]


def test_lviba_lidar_edge_derivatives(oracle, synthetic):
    """EdgeLidar of LocalLVIBA (LidarCovisRes::ComputeJandH, SF/src/LidarRes.cc:89-128): the translation rows are the gradient of
    r with respect to ImuCamPose::Update's increment; the rotation rows carry the reference's extra InverseRightJacobianSO3
    factor -- removing it (J_w' = Rlb^T A^-T ...) gives the finite-difference gradient, which pins the restatement."""
    w = synthetic.inertial_window(0, n_opt=6, n_points=300)
    K = len(w["kf33"])
    win = list(range(K - 1, K - 7, -1))
    clouds = synthetic.inertial_window_clouds(w, win, n_points=2400)
    tbl = synthetic.tbl7()
    n, err, J, H = oracle.lidar_window_evaluate_body(w["kf33"], w["kf33"], win, clouds, synthetic.TCL7, tbl)
    assert n > 50 and err > 0
    assert np.allclose(H, H.T, rtol=1e-9, atol=1e-9 * np.abs(H).max())

    def hat(v):
        return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])

    from scipy.spatial.transform import Rotation
    Rcl = synthetic._quat_R(synthetic.TCL7[:4])
    Rlb = synthetic._quat_R(tbl[:4]).T
    t_bl = tbl[4:].astype(np.float64)
    for i, k in enumerate(win[:3]):
        g = np.zeros(6)
        for c in range(6):
            h = 2e-3 if c < 3 else 2e-2
            r = []
            for sgn in (1, -1):
                u = np.zeros(6); u[c] = sgn * h
                kf = w["kf33"].copy()
                kf[k], _ = oracle.imu_pose_update(kf[k], 0, w["calib24"], u)
                e = oracle.lidar_window_evaluate_body(w["kf33"], kf, win, clouds, synthetic.TCL7, tbl, derivatives=False)[1]
                r.append(e * e)
            g[c] = (r[0] - r[1]) / (2 * h)
        jw, jt = J[6 * i:6 * i + 3], J[6 * i + 3:6 * i + 6]
        assert np.allclose(jt, g[3:], rtol=2e-2, atol=2e-2 * np.abs(g).max())
        Rwl = w["kf33"][k][:9].reshape(3, 3).T @ Rcl
        rwl = Rotation.from_matrix(Rwl).as_rotvec()
        th = np.linalg.norm(rwl)
        Wm = hat(rwl)
        Jrinv = np.eye(3) + 0.5 * Wm + (1 / th ** 2 - (1 + np.cos(th)) / (2 * th * np.sin(th))) * Wm @ Wm
        Rwb = Rwl @ Rlb
        A, B = (Jrinv @ Rlb).T, (Rwb @ hat(t_bl)).T
        Jt = Rwb @ jt
        Jw = np.linalg.solve(A, jw + B @ Jt)
        assert np.allclose(Rlb.T @ Jw - B @ Jt, g[:3], rtol=2e-2, atol=2e-2 * np.abs(g).max())


def test_lviba_runs_and_reports(oracle, synthetic):
    w = synthetic.inertial_window(1, n_opt=7, n_points=300)
    K = len(w["kf33"])
    win = list(range(K - 1, K - 7, -1))
    clouds = synthetic.inertial_window_clouds(w, win, n_points=1800)
    pre = []
    for s, t1, t2 in w["samples"]:
        _, f = oracle.imu_preintegrate(s, t1, t2, w["bias6"], *synthetic.IMU_NOISE)
        pre.append(oracle.pack_preintegrated(f, w["bias6"]))
    r = oracle.local_lviba(w["kf33"], w["fixed"], w["has_imu"], w["calib24"], w["points"], w["edges"], w["link4"], np.stack(pre), w["cam"], win, clouds,
                           synthetic.TCL7, synthetic.tbl7(), 1.0)
    assert r[4] >= 3 and r[7] > 50 and r[6][1] < 0.01 * r[6][0]
    err0 = np.linalg.norm(w["kf33"][:, 21:24] - w["kf33_true"][:, 21:24], axis=1).mean()
    err1 = np.linalg.norm(r[0][:, 21:24] - w["kf33_true"][:, 21:24], axis=1).mean()
    assert err1 < 0.5 * err0
    assert np.array_equal(r[0][0], w["kf33"][0])


# ---- Optimizer::PoseInertialOptimizationLastKeyFrame / LastFrame (row a10') ----------------------------------------------------------
def _pose_inertial_problem(oracle, synthetic, seed, last_frame, **kw):
    w = synthetic.pose_inertial_problem(seed, last_frame=last_frame, **kw)
    f = oracle.imu_preintegrate(w["samples"], w["t1"], w["t2"], w["bias6"], *synthetic.IMU_NOISE)
    w["pre298"] = oracle.pack_preintegrated(f[1] if isinstance(f, tuple) else f, w["bias6"])
    return w


@pytest.mark.parametrize("last_frame", [False, True])
@pytest.mark.parametrize("seed", [0, 1])
def test_pose_inertial_optimisation(oracle, synthetic, seed, last_frame):
    w = _pose_inertial_problem(oracle, synthetic, seed, last_frame)
    cur, oth, outlier, prior, rv, (n0, nbad, ninl) = oracle.pose_inertial(w["cur33"], w["other33"], last_frame, w["prior246"], w["calib24"], w["pre298"],
                                                                          w["pre298"], w["Xw"], w["edges"], w["close"], w["cam"])
    E = len(w["edges"])
    assert n0 == E and rv == n0 - nbad and ninl + nbad == E and ninl > 0.8 * E
    # the gross outliers are found, few good edges are lost
    assert outlier[w["gross"]].mean() > 0.9 and outlier[~w["gross"]].mean() < 0.08
    # the pose moves towards the truth
    err0 = np.linalg.norm(w["cur33"][21:24] - w["cur33_true"][21:24])
    err1 = np.linalg.norm(cur[21:24] - w["cur33_true"][21:24])
    assert err1 < 0.4 * err0 and err1 < 0.03
    R0, R1, Rt = (x[12:21].reshape(3, 3) for x in (w["cur33"], cur, w["cur33_true"]))
    ang = lambda R: np.degrees(np.arccos(np.clip((np.trace(R @ Rt.T) - 1) / 2, -1, 1)))
    assert ang(R1) < 0.3 * ang(R0)
    # camera pose consistent with the body pose (ImuCamPose::Update)
    Rcb, tcb = w["calib24"][:9].reshape(3, 3), w["calib24"][9:12]
    assert np.allclose(cur[:9].reshape(3, 3), Rcb @ R1.T, atol=1e-12) and np.allclose(cur[9:12], Rcb @ (-R1.T @ cur[21:24]) + tcb, atol=1e-10)
    # the other state: fixed for the keyframe form, moved by the previous-frame form
    assert np.array_equal(oth, w["other33"]) != last_frame
    # the new prior: the frame's state and a symmetric positive semi-definite information matrix
    H = prior[21:].reshape(15, 15)
    assert np.array_equal(prior[:9], cur[12:21]) and np.array_equal(prior[9:12], cur[21:24]) and np.array_equal(prior[12:15], cur[24:27])
    assert np.allclose(H, H.T, rtol=1e-9, atol=1e-6 * np.abs(H).max())
    assert np.linalg.eigvalsh((H + H.T) / 2).min() > -1e-6 * np.abs(H).max() and np.linalg.eigvalsh((H + H.T) / 2).max() > 1.0
    # pose block: at least the sum of the inlier edges' J^T W J (a few thousand per pixel^-2 unit)
    assert H[0, 0] > 100 and H[3, 3] > 100


def test_pose_inertial_recovery_pass(oracle, synthetic):
    """Fewer than 30 inliers and !bRecInit: every edge is re-tested against 18 / 24 (Optimizer.cc:2738-2765); with bRecInit it is not."""
    w = _pose_inertial_problem(oracle, synthetic, 3, False, n_points=30, outlier_frac=0.3)
    a = oracle.pose_inertial(w["cur33"], w["other33"], False, None, w["calib24"], w["pre298"], w["pre298"], w["Xw"], w["edges"], w["close"], w["cam"], rec_init=False)
    b = oracle.pose_inertial(w["cur33"], w["other33"], False, None, w["calib24"], w["pre298"], w["pre298"], w["Xw"], w["edges"], w["close"], w["cam"], rec_init=True)
    assert a[5][2] < 30 and np.allclose(a[0], b[0])      # same optimisation
    assert a[2].sum() <= b[2].sum() and a[4] >= b[4]      # the recovery can only clear flags
