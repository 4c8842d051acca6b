"""The iterated error-state Kalman filter of the LiDAR-inertial front end:
 - CPU: properties of the oracle's restatement (charts, predict, update) and the product's host-only predict / forward propagation
   with covariance against the oracle;
 - GPU: tc2li_lidar_eskf_update (neighbour search, plane fit, selection and the normal equations of the measurement rows on the
   device, 23 x 23 algebra on the host) against the oracle: same number of model evaluations / searches / selected points,
   state within 1e-4 relative (BASELINE.json's bar for poses), covariance within 1e-6 relative."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

G = 9.809


def mkstate(R, p, v=(0, 0, 0), bg=(0, 0, 0), ba=(0, 0, 0), g=(0, 0, -G), oR=np.eye(3), oT=(0, 0, 0)):
    return np.concatenate([p, np.ravel(R), v, bg, ba, g, np.ravel(oR), oT]).astype(np.float64)


def scene_problem(oracle, synthetic, pkg, frame=1, every=1):
    scene = synthetic.Scene(1)
    down0 = oracle.voxel_grid(oracle.lidar_preprocess(synthetic.lidar_scan(scene, 0)))
    down1 = oracle.voxel_grid(oracle.lidar_preprocess(synthetic.lidar_scan(scene, frame)))[::every]
    st0 = pkg.capi.pack_lidar_state(*synthetic.lidar_state(0)[:2])
    boot = oracle.KdTree(down0[:8])
    world0 = oracle.feature_extraction(boot, down0, st0)["world"]
    R1, t1 = synthetic.sensor_pose(frame)
    return world0, down1, mkstate(R1, t1)


# ---- CPU ----------------------------------------------------------------------------------------------------------------------
def test_s2_chart_and_boxplus_roundtrip(oracle):
    rng = np.random.default_rng(0)
    for _ in range(5):
        g = rng.normal(0, 1, 3); g *= G / np.linalg.norm(g)
        Bx, Nx, Mx = oracle.s2_matrices(g)
        assert np.allclose(Bx.T @ g, 0, atol=1e-12) and np.allclose(Bx.T @ Bx, np.eye(2), atol=1e-12)
        assert np.allclose(Nx @ Mx, np.eye(2), atol=1e-12)
        x = mkstate(Rotation.from_rotvec(rng.normal(0, 0.5, 3)).as_matrix(), rng.normal(0, 5, 3), v=rng.normal(0, 3, 3), g=g,
                    oR=Rotation.from_rotvec(rng.normal(0, 0.1, 3)).as_matrix(), oT=rng.normal(0, 0.1, 3))
        d = rng.normal(0, 0.05, 23)
        y = oracle.eskf_boxplus(x, d)
        assert np.allclose(oracle.eskf_boxminus(y, x), d, atol=1e-12)
        assert abs(np.linalg.norm(y[21:24]) - G) < 1e-12  # gravity stays on the sphere
    # chart singularity (vec[0] = -length): the fixed fallback of S2_Bx
    Bx, _, _ = oracle.s2_matrices(np.array([-G, 0.0, 0.0]))
    assert np.array_equal(Bx, np.array([[0, 0], [0, -1.0], [1.0, 0]]))


def test_predict_covariance_properties(oracle):
    rng = np.random.default_rng(1)
    x = mkstate(Rotation.from_rotvec([0.1, 0.2, -0.3]).as_matrix(), [1.0, 2, 3], v=[5.0, 0.3, -0.1], bg=[1e-3, 0, -1e-3], ba=[0.02, -0.01, 0])
    A = rng.normal(0, 1, (23, 23))
    P = A @ A.T * 1e-4 + np.eye(23) * 1e-4
    Q = np.diag([0.1] * 6 + [1e-4] * 6)
    acc, gyr, dt = np.array([0.2, -0.1, 9.7]), np.array([0.02, -0.01, 0.3]), 0.01
    x2, P2 = oracle.eskf_predict(x, P, Q, acc, gyr, dt)
    assert np.allclose(P2, P2.T, atol=1e-15) and np.linalg.eigvalsh(P2).min() > 0
    # state: x oplus f dt
    R = x[3:12].reshape(3, 3)
    assert np.allclose(x2[:3], x[:3] + x[12:15] * dt)
    assert np.allclose(x2[3:12].reshape(3, 3), R @ Rotation.from_rotvec((gyr - x[15:18]) * dt).as_matrix(), atol=1e-13)
    assert np.allclose(x2[12:15], x[12:15] + (R @ (acc - x[18:21]) + x[21:24]) * dt)
    assert np.array_equal(x2[15:], x[15:])  # biases, gravity, extrinsics
    # position block: F = [I .. dt I ..]
    Ppp = P[:3, :3] + dt * (P[:3, 12:15] + P[12:15, :3]) + dt * dt * P[12:15, 12:15]
    assert np.allclose(P2[:3, :3], Ppp, rtol=1e-12)
    # bias blocks only gain the random-walk noise
    assert np.allclose(P2[15:18, 15:18], P[15:18, 15:18] + dt * dt * Q[6:9, 6:9], rtol=1e-12)
    # dt = 0 leaves everything as it was
    x3, P3 = oracle.eskf_predict(x, P, Q, acc, gyr, 0.0)
    assert np.allclose(x3, x, atol=1e-15) and np.allclose(P3, P, rtol=1e-13)


def test_product_predict_matches_the_oracle(pkg, oracle):
    """tc2li_eskf_predict / tc2li_lidar_imu_propagate_cov are host-only (23 x 23 algebra per IMU sample): no GPU needed."""
    rng = np.random.default_rng(2)
    for seed in range(3):
        x = mkstate(Rotation.from_rotvec(rng.normal(0, 0.4, 3)).as_matrix(), rng.normal(0, 3, 3), v=[8, 0.2, 0], bg=[1e-3, 0, -1e-3], ba=[0.01, 0.02, 0],
                    g=[0.05, -0.02, -9.8089], oR=Rotation.from_rotvec(rng.normal(0, 0.05, 3)).as_matrix(), oT=[0.1, 0, 0.05])
        x[21:24] *= G / np.linalg.norm(x[21:24])
        A = rng.normal(0, 1, (23, 23))
        P = A @ A.T * 1e-5 + np.eye(23) * 1e-4
        Q = np.diag(rng.uniform(1e-4, 0.1, 12))
        acc, gyr = [0.1, -0.2, 9.8] + rng.normal(0, 0.05, 3), [0.01, 0.02, 0.3] + rng.normal(0, 0.01, 3)
        ws, wP = oracle.eskf_predict(x, P, Q, acc, gyr, 0.01)
        gs, gP = pkg.capi.eskf_predict(x, P, Q, acc, gyr, 0.01)
        assert np.allclose(gs, ws, rtol=1e-13, atol=1e-13)
        assert np.allclose(gP, wP.reshape(23, 23), rtol=1e-11, atol=1e-16)
        t = np.arange(995, 1012) / 100.0
        imu = np.zeros((len(t), 7)); imu[:, 0] = t
        imu[:, 1:4] = [0.1, -0.2, 9.8] + rng.normal(0, 0.05, (len(t), 3))
        imu[:, 4:7] = [0.01, 0.02, 0.3] + rng.normal(0, 0.01, (len(t), 3))
        cov12 = np.array([0.1] * 6 + [1e-4] * 6)
        last6 = rng.normal(0, 0.1, 6)
        ws, wP, wp = oracle.imu_propagate_cov(x, P, cov12, imu, 10.0, 10.1, 9.999, 1.002, last6)
        gs, gP, gp, _ = pkg.capi.lidar_imu_propagate_cov(x, P, cov12, imu, 10.0, 10.1, 9.999, 1.002, last6)
        assert len(gp) == len(wp) >= 10
        assert np.allclose(gs, ws, rtol=1e-12, atol=1e-12) and np.allclose(gp, wp, rtol=1e-12, atol=1e-12)
        assert np.allclose(gP, wP.reshape(23, 23), rtol=1e-9, atol=1e-15)
        # the state part agrees with the covariance-free propagation
        s0, p0, _ = pkg.capi.lidar_imu_propagate(x, imu, 10.0, 10.1, 9.999, 1.002, last6)
        assert np.allclose(s0, gs, rtol=1e-12, atol=1e-12)
        assert np.linalg.eigvalsh((gP + gP.T) / 2).min() > 0


def test_oracle_update_pulls_the_state_to_the_map(oracle, synthetic, pkg):
    world0, down1, xt = scene_problem(oracle, synthetic, pkg, every=2)
    tree = oracle.KdTree(world0)
    d0 = np.concatenate([[0.1, -0.12, 0.06], [0.01, -0.015, 0.02], np.zeros(17)])
    xe = oracle.eskf_boxplus(xt, d0)
    P = np.eye(23) * 1e-2
    xs, Ps, info = oracle.eskf_update(xe, P, tree, down1)
    assert info["finished"] and info["effct_feat_num"] > 1000 and 1 <= info["searches"] <= info["calls"] <= 5
    e0, e1 = oracle.eskf_boxminus(xe, xt), oracle.eskf_boxminus(xs, xt)
    # lateral / vertical position and attitude are observable in the corridor scene; along-track is not
    assert np.abs(e1[1:3]).max() < 0.1 * np.abs(e0[1:3]).max() and np.abs(e1[3:6]).max() < 0.1 * np.abs(e0[3:6]).max()
    assert np.allclose(Ps, Ps.reshape(23, 23).T.reshape(Ps.shape), atol=1e-12)
    Pm = Ps.reshape(23, 23)
    assert Pm[1, 1] < 1e-2 * P[1, 1] and Pm[4, 4] < 1e-2 * P[4, 4]
    # no effective points: state and covariance stay as propagated
    far = xe.copy(); far[:3] += 500.0
    xs2, Ps2, info2 = oracle.eskf_update(far, P, tree, down1)
    assert not info2["finished"] and info2["effct_feat_num"] == 0 and np.array_equal(xs2, far) and np.allclose(Ps2.reshape(23, 23), P)


# ---- GPU ----------------------------------------------------------------------------------------------------------------------
def _compare_update(pkg, oracle, world0, body, xe, P, **kw):
    tree = oracle.KdTree(world0)
    want_x, want_P, info = oracle.eskf_update(xe, P, tree, body, **kw)
    fe = pkg.LidarFrontEnd(max_points_per_scan=max(len(body), 256), max_scans=1)
    m = pkg.LidarMap(); m.Build(world0)
    got_x, got_P, st = fe.eskf_update(m, body, xe, P, **kw)
    assert (st.calls, st.searches, st.converged, bool(st.finished)) == (info["calls"], info["searches"], info["converged"], info["finished"])
    assert st.effct_feat_num == info["effct_feat_num"]
    assert abs(st.res_mean_last - info["res_mean_last"]) <= 1e-5 * max(info["res_mean_last"], 1e-9)
    scale = max(1.0, np.abs(want_x[:3]).max())
    assert np.abs(got_x[:3] - want_x[:3]).max() / scale < 1e-4
    assert np.abs(got_x[3:12] - want_x[3:12]).max() < 1e-6 and np.abs(got_x[24:33] - want_x[24:33]).max() < 1e-6  # rotations
    assert np.allclose(got_x[12:24], want_x[12:24], rtol=1e-4, atol=1e-6) and np.allclose(got_x[33:], want_x[33:], rtol=1e-4, atol=1e-6)
    wP = want_P.reshape(23, 23)
    assert np.abs(got_P - wP).max() <= 1e-6 * np.abs(wP).max()
    return got_x, got_P, st, fe, m


@pytest.mark.gpu
@pytest.mark.parametrize("seed,ext,max_iter", [(0, False, 4), (1, True, 4), (2, False, 2), (3, True, 6)])
def test_eskf_update_matches_the_oracle(pkg, oracle, synthetic, seed, ext, max_iter):
    world0, down1, xt = scene_problem(oracle, synthetic, pkg)
    rng = np.random.default_rng(seed)
    d0 = np.concatenate([rng.normal(0, 0.08, 3), rng.normal(0, 0.01, 3), rng.normal(0, 0.002, 3) * ext, rng.normal(0, 0.01, 3) * ext,
                         rng.normal(0, 0.05, 3), np.zeros(8)])
    xe = oracle.eskf_boxplus(xt, d0)
    A = rng.normal(0, 1, (23, 23))
    P = A @ A.T * 1e-5 + np.diag([1e-2] * 3 + [1e-3] * 3 + [1e-5] * 6 + [1e-2] * 3 + [1e-4] * 6 + [1e-5] * 2)
    got_x, got_P, st, fe, m = _compare_update(pkg, oracle, world0, down1, xe, P, max_iter=max_iter, extrinsic_est_en=ext)
    assert st.effct_feat_num > 3000 and st.finished
    # map_incremental afterwards uses the neighbours of the last evaluation, as the reference does
    n_map, n_add, n_noneed = m.map_incremental(fe, 0, np.concatenate([got_x[3:12], got_x[:3], got_x[24:33], got_x[33:36]]))
    assert n_add + n_noneed > 0 and n_map >= len(world0)


@pytest.mark.gpu
def test_eskf_update_edge_cases(pkg, oracle, synthetic):
    world0, down1, xt = scene_problem(oracle, synthetic, pkg)
    P = np.eye(23) * 1e-3
    # fewer effective points than states: the gain goes through the innovation form (esekfom.hpp:1704-1727)
    few = down1[::400][:20]
    _, _, st, _, _ = _compare_update(pkg, oracle, world0, few, xt, P)
    assert 1 <= st.effct_feat_num < 23
    # no effective points at all: nothing changes
    far = xt.copy(); far[:3] += 500.0
    got_x, got_P, st, _, _ = _compare_update(pkg, oracle, world0, down1[::10], far, P)
    assert st.effct_feat_num == 0 and not st.finished and np.array_equal(got_x, far) and np.allclose(got_P, P)
    # tight limits: never converges, runs all iterations with a single neighbour search
    _, _, st, _, _ = _compare_update(pkg, oracle, world0, down1[::3], oracle.eskf_boxplus(xt, np.concatenate([[0.05, 0.05, 0.02], np.zeros(20)])), P,
                                     limit=np.full(23, 1e-12), max_iter=3)
    assert st.calls == 4 and st.converged == 0 and st.searches == 2  # the second-to-last iteration forces converge = true (:1815-1818)
    # empty scan and invalid arguments
    fe = pkg.LidarFrontEnd(max_points_per_scan=1024, max_scans=1)
    m = pkg.LidarMap(); m.Build(world0)
    x, Pm, st = fe.eskf_update(m, down1[:0], xt, P)
    assert st.calls == 0 and np.array_equal(x, xt)
    with pytest.raises(pkg.capi.Tc2liError):
        fe.eskf_update(m, down1[:10], xt, P, R=0.0)
