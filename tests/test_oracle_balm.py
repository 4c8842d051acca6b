"""CPU tests of the BALM (LiDAR plane) oracle: PointCluster::transform against brute-force point transforms
(SURVEY.md section 4), the 3x3 eigen solver against numpy, the analytic Jacobian / Hessian of sum(N * lambda_min)
against finite differences of the residual, the quirky edge inside the LM loop."""
import numpy as np
import pytest


def rot(rv):
    th = np.linalg.norm(rv)
    if th < 1e-12:
        return np.eye(3)
    k = rv / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


@pytest.fixture(scope="module")
def scene():
    """Two planes seen from 4 LiDAR poses."""
    rng = np.random.default_rng(0)
    W = 4
    Twl, clouds = [], []
    for i in range(W):
        R = rot(rng.normal(0, 0.02, 3))
        p = np.array([0.8 * i, 0.05 * i, 0.02 * i]) + rng.normal(0, 0.01, 3)
        n = 1200
        ground = np.stack([rng.uniform(2, 8, n) + p[0], rng.uniform(-3, 3, n), np.full(n, -1.7)], 1)
        wall = np.stack([rng.uniform(2, 8, n) + p[0], np.full(n, 4.0), rng.uniform(-1.7, 1.0, n)], 1)
        Xw = np.concatenate([ground, wall]) + rng.normal(0, 0.02, (2 * n, 3))
        clouds.append(((Xw - p) @ R).astype(np.float32))  # R^T (Xw - p)
        Twl.append(np.concatenate([R.reshape(-1), p]))
    return np.array(Twl), clouds


def test_balm_jacobian_against_finite_differences(oracle, scene):
    Twl, clouds = scene
    W = len(Twl)
    n_planes, res, J, H, _ = oracle.balm_evaluate(Twl, clouds)
    assert n_planes > 20 and res > 0
    d = 2e-3
    evals = []
    for i in range(W):
        for k in range(6):
            for sgn in (+1, -1):
                T = Twl.copy()
                R = T[i, :9].reshape(3, 3)
                if k < 3:
                    e = np.zeros(3); e[k] = sgn * d
                    T[i, :9] = (R @ rot(e)).reshape(-1)  # IMUST::operator+= : R <- R Exp(dtheta)
                else:
                    T[i, 9 + k - 3] += sgn * d
                evals.append(T)
    er = oracle.balm_evaluate(Twl, clouds, np.array(evals))[4]
    fd = (er[0::2] - er[1::2]) / (2 * d)
    # the residual is evaluated at poses rounded through float (LidarRes.cc:221-235), which limits the FD accuracy
    assert np.allclose(fd, J, rtol=0.03, atol=0.02 * np.abs(J).max())
    # second differences along single coordinates against the Hessian diagonal
    dd = (er[0::2] - 2 * res + er[1::2]) / (d * d)
    big = np.abs(np.diag(H)) > 0.05 * np.abs(np.diag(H)).max()
    assert np.allclose(dd[big], np.diag(H)[big], rtol=0.15)
    assert np.allclose(H, H.T, rtol=1e-9, atol=1e-9 * np.abs(H).max())


def test_residual_is_invariant_to_a_common_rigid_motion(oracle, scene):
    Twl, clouds = scene
    G = rot(np.array([0.3, -0.2, 0.5])); g = np.array([5.0, -2.0, 1.0])
    moved = Twl.copy()
    for i in range(len(Twl)):
        R, p = Twl[i, :9].reshape(3, 3), Twl[i, 9:]
        moved[i, :9] = (G @ R).reshape(-1); moved[i, 9:] = G @ p + g
    _, res, _, _, er = oracle.balm_evaluate(Twl, clouds, moved[None])
    assert abs(er[0] - res) < 2e-3 * res  # lambda_min of the merged clusters does not depend on the common frame


def test_lidar_edge_in_the_lm_loop(oracle, synthetic):
    w = synthetic.ba_window(0, n_opt=6, n_fix=6, n_points=600, pose_noise=(0.1, 0.01))
    win = [11, 10, 9, 8]
    clouds = synthetic.ba_window_clouds(w, win, n_points=2400)
    base = oracle.local_ba(w["poses"], w["fixed"], w["points"], w["edges"], w["cam"])
    r = oracle.local_ba_lidar(w["poses"], w["fixed"], w["points"], w["edges"], w["cam"], win, clouds, synthetic.TCL7, 1.0)
    assert r[6] > 30                                        # planes found
    assert r[4] >= 1 and np.all(np.diff(r[5]["chi2"]) <= 1e-9)
    assert abs(r[5]["chi2"][-1] - base[5]["chi2"][-1]) > 1e-6   # the edge takes part in the cost
    assert np.abs(r[0] - base[0]).max() > 1e-7                  # and moves the poses
    assert np.array_equal(r[0][w["fixed"] > 0], w["poses"][w["fixed"] > 0])
    W = len(win)
    assert r[7]["JacT"].shape == (6 * W,) and np.abs(r[7]["Hessian"]).max() > 0
    # window of 2 keyframes is below the reference's threshold (> 2, OptimizerWithLidar.cc:236) but the edge itself works
    r2 = oracle.local_ba_lidar(w["poses"], w["fixed"], w["points"], w["edges"], w["cam"], win[:3], clouds[:3], synthetic.TCL7, 0.01)
    assert r2[4] >= 1


def _canon(P6v3n, coe):
    """Order-independent view of a plane list: sort planes by a key built from their content."""
    key = np.round(np.concatenate([coe[:, None], P6v3n[:, :, 6:9].sum(1)], 1), 6)
    order = np.lexsort(key.T[::-1])
    return P6v3n[order], coe[order]


def test_host_plane_extraction_matches_the_oracle(pkg, oracle, synthetic):
    """tc2li_host_lidar_planes (the product's host stage; no GPU needed) against cut_voxel / recut / tras_opt of the oracle:
    same planes, bit-identical cluster sums (the visiting order of planes is unspecified in the reference: unordered_map)."""
    for seed, n_pts, noise in [(0, 2400, (0.1, 0.01)), (3, 1500, (0.02, 0.002))]:
        w = synthetic.ba_window(seed, n_opt=6, n_fix=6, n_points=300, pose_noise=noise)
        win = [11, 10, 9, 8, 7]
        clouds = synthetic.ba_window_clouds(w, win, n_points=n_pts)
        oc, ocoe = oracle.lidar_planes(w["poses"], win, clouds, synthetic.TCL7)
        pc, pcoe = pkg.capi.lidar_planes_host(w["poses"], win, clouds, synthetic.TCL7)
        assert len(ocoe) == len(pcoe) and len(ocoe) > 20
        o10 = np.concatenate([oc[:, :, [0, 1, 2, 4, 5, 8]], oc[:, :, 9:13]], 2)
        assert np.array_equal(oc[:, :, [1, 2, 5]], oc[:, :, [3, 6, 7]])  # P is symmetric
        a, ac = _canon(o10, ocoe)
        b, bc = _canon(pc, pcoe)
        assert np.array_equal(ac, bc)
        assert np.array_equal(a, b)


def test_host_plane_extraction_rejects_bad_windows(pkg, synthetic):
    w = synthetic.ba_window(0, n_opt=3, n_fix=3, n_points=100)
    clouds = synthetic.ba_window_clouds(w, [5, 4], n_points=200)
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.lidar_planes_host(w["poses"], [5, 99], clouds, synthetic.TCL7)
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.lidar_planes_host(w["poses"], [5, 4], [clouds[0], np.zeros((0, 3), np.float32)], synthetic.TCL7)
