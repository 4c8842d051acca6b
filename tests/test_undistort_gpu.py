"""GPU parity of the scan motion compensation (ImuProcess::UndistortPcl) with the oracle.  The
point order (std::sort on the time offsets, ties included) must be identical; coordinates are computed in double with the
device's sin / cos, so they are compared to one float ulp and the number of differing values is reported."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def imu_stream(t0, t1, rate=100.0, seed=0):
    rng = np.random.default_rng(seed)
    k0, k1 = int(np.floor(t0 * rate)), int(np.ceil(t1 * rate))
    t = np.arange(k0, k1 + 1) / rate
    imu = np.zeros((len(t), 7))
    imu[:, 0] = t
    ph = 2 * np.pi * 0.5 * t
    imu[:, 1:4] = np.stack([0.4 * np.sin(ph), 0.2 * np.cos(ph), 9.81 + 0.1 * np.sin(2 * ph)], 1) + rng.normal(0, 0.02, (len(t), 3))
    imu[:, 4:7] = np.stack([0.02 * np.sin(ph), 0.03 * np.cos(ph), 0.25 + 0.05 * np.sin(ph)], 1) + rng.normal(0, 0.002, (len(t), 3))
    return imu


def initial_state(seed=0):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(seed)
    R = Rotation.from_rotvec(rng.normal(0, 0.2, 3)).as_matrix()
    Rli = Rotation.from_rotvec([0.01, -0.02, 0.015]).as_matrix()
    st = np.concatenate([rng.normal(0, 5, 3), R.reshape(-1), [9.0, 0.5, -0.1], [1e-3, -2e-3, 5e-4], [0.02, 0.01, -0.03], [0, 0, -9.81],
                         Rli.reshape(-1), [0.05, -0.02, 0.1]])
    return st


def lidar_state24(st36):
    return np.concatenate([st36[3:12], st36[0:3], st36[24:33], st36[33:36]])


@pytest.fixture(scope="module")
def fe(pkg):
    return pkg.LidarFrontEnd(max_points_per_scan=140000, max_scans=1)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_propagate_and_undistort(pkg, fe, oracle, synthetic, seed):
    scan = synthetic.lidar_scan(synthetic.Scene(seed), seed + 1)
    pts = oracle.lidar_preprocess(scan)            # curvature = time offset in ms over the 0.1 s sweep
    beg, end = 10.0 + 0.1 * seed, 10.1 + 0.1 * seed
    imu = imu_stream(beg - 0.012, end + 0.004, seed=seed)
    st0 = initial_state(seed)
    last6 = np.array([0.1, -0.05, 0.02, 0.01, 0.0, 0.24])
    want_st, want_poses = oracle.imu_propagate(st0, imu, beg, end, beg - 0.001, 9.81 / 9.79, last6)
    got_st, got_poses, _ = pkg.capi.lidar_imu_propagate(st0, imu, beg, end, beg - 0.001, 9.81 / 9.79, last6)
    assert len(got_poses) == len(want_poses) >= 10
    assert np.allclose(got_poses, want_poses, rtol=1e-12, atol=1e-12) and np.allclose(got_st, want_st, rtol=1e-12, atol=1e-12)
    want = oracle.undistort(pts, want_poses, lidar_state24(want_st))
    got = fe.undistort(pts, want_poses, lidar_state24(want_st))
    assert len(got) == len(want) == len(pts)
    # identical order: every other field travels with the point
    for name in ("intensity", "curvature", "normal_x", "pad0"):
        assert np.array_equal(got[name], want[name]), name
    assert np.all(np.diff(got["curvature"]) >= 0)
    xyz_g = np.stack([got["x"], got["y"], got["z"]], 1); xyz_w = np.stack([want["x"], want["y"], want["z"]], 1)
    ulp = np.spacing(np.abs(xyz_w).astype(np.float32))
    assert np.all(np.abs(xyz_g - xyz_w) <= ulp)
    assert (xyz_g != xyz_w).mean() < 1e-3          # in practice 0: double results round to the same float
    # the compensation really moves points (several cm at 9 m/s over 0.1 s) except those at the very end of the sweep
    moved = np.linalg.norm(xyz_g - np.stack([pts["x"], pts["y"], pts["z"]], 1)[np.argsort(pts["curvature"], kind="stable")], axis=1)
    assert np.median(moved) > 0.05


def test_undistort_edge_cases(pkg, fe, oracle, synthetic):
    scan = synthetic.lidar_scan(synthetic.Scene(5), 2)
    pts = oracle.lidar_preprocess(scan)[:5000]
    st0 = initial_state(3)
    imu = imu_stream(19.99, 20.11, seed=3)
    st, poses = oracle.imu_propagate(st0, imu, 20.0, 20.1, 19.999, 1.0, np.zeros(6))
    s24 = lidar_state24(st)
    # points at time 0 are not compensated; the first point of the sorted scan is compensated once per earlier interval
    p2 = pts.copy(); p2["curvature"][:7] = 0.0; p2["curvature"][7] = 55.0
    for p in (p2, pts[:1], pts[:2], pts[:0]):
        want = oracle.undistort(p, poses, s24)
        got = fe.undistort(p, poses, s24)
        assert len(got) == len(want)
        for name in ("x", "y", "z"):
            assert np.all(np.abs(got[name] - want[name]) <= np.spacing(np.abs(want[name]))), name
        assert np.array_equal(got["curvature"], want["curvature"]) and np.array_equal(got["intensity"], want["intensity"])
    first = pts[:3].copy(); first["curvature"] = [80.0, 90.0, 95.0]   # the first sorted point lies late in the sweep
    want = oracle.undistort(first, poses, s24); got = fe.undistort(first, poses, s24)
    for name in ("x", "y", "z"):
        assert np.all(np.abs(got[name] - want[name]) <= np.spacing(np.abs(want[name])))
    # a single pose: nothing to do but the sort
    got = fe.undistort(pts, poses[:1], s24); want = oracle.undistort(pts, poses[:1], s24)
    assert np.array_equal(got, want)
    with pytest.raises(pkg.capi.Tc2liError):
        fe.undistort(pts, np.zeros((65, 22)), s24)
