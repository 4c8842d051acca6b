"""Row b4: the pose plumbing between the camera thread and the LiDAR front end -- UpdateLidarPose, InterpolateSE3, the
transform chains of Tracking::SyncWithLidar / BuildLidarFeat4KeyFrame and LidarFrontEndTools::transformPointCloud.
CPU: the oracle's float Sophus / Eigen restatement against a float64 scipy statement (pins the oracle), and the product's host entries
(no device needed) against the oracle bit for bit.  GPU: the transform kernel (host arrays and the batched, device-resident form)."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation, Slerp


def rand_pose(rng, angle=1.0, trans=5.0):
    q = Rotation.from_rotvec(rng.normal(0, angle, 3)).as_quat()
    return np.concatenate([q, rng.normal(0, trans, 3)]).astype(np.float32)


def mat(p7):
    M = np.eye(4)
    M[:3, :3] = Rotation.from_quat(np.asarray(p7[:4], np.float64)).as_matrix()
    M[:3, 3] = p7[4:]
    return M


def close(p7, M, tol=2e-5):
    return np.abs(mat(p7) - M).max() < tol * max(1.0, np.abs(M).max())


def test_oracle_se3f_against_float64(oracle):
    rng = np.random.default_rng(0)
    for k in range(40):
        a, b = rand_pose(rng, 1.2 if k % 4 else 1e-4), rand_pose(rng)
        t = float(rng.uniform(0, 1))
        r = oracle.se3f_ops(a, b, t)
        A, B = mat(a), mat(b)
        assert close(r["inverse"], np.linalg.inv(A)) and close(r["mul"], A @ B)
        # exp(log(T)) = T; for tiny angles Sophus' float V^-1 = I - Omega/2 + (1 - theta cos(theta/2) / (2 sin(theta/2))) / theta^2 Omega^2 cancels
        # catastrophically above its 1e-5 switch to the series (the reference carries that error too)
        assert close(r["exp_log"], A, 2e-5 if k % 4 else 2e-3)
        w = r["log"][3:].astype(np.float64)
        assert np.abs(Rotation.from_rotvec(w).as_matrix() - A[:3, :3]).max() < 2e-5
        # InterpolateSE3: slerp of the rotations, lerp of the translations
        Ri = Slerp([0, 1], Rotation.from_matrix(np.stack([A[:3, :3], B[:3, :3]])))([t]).as_matrix()[0]
        Mi = np.eye(4); Mi[:3, :3] = Ri; Mi[:3, 3] = A[:3, 3] + t * (B[:3, 3] - A[:3, 3])
        assert close(r["interpolate"], Mi, 5e-5)


def _scenario(rng, synthetic):
    Tcl = np.asarray(synthetic.TCL7, np.float32)
    Tlc = np.zeros(7, np.float32)
    Minv = np.linalg.inv(mat(Tcl))
    Tlc[:4], Tlc[4:] = Rotation.from_matrix(Minv[:3, :3]).as_quat(), Minv[:3, 3]
    Tcw_last, vel = rand_pose(rng, 0.3, 20.0), rand_pose(rng, 0.02, 0.5)
    Mcur = mat(vel) @ mat(Tcw_last)
    Tcw_cur = np.concatenate([Rotation.from_matrix(Mcur[:3, :3]).as_quat(), Mcur[:3, 3]]).astype(np.float32)
    return Tcl, Tlc, Tcw_last, Tcw_cur, vel


def test_update_lidar_pose_and_chains(pkg, oracle, synthetic):
    """Product host entries == oracle bit for bit; oracle == float64 statement of LidarFrontEnd.cpp:786-800 / Tracking.cc:1510-1630."""
    rng = np.random.default_rng(1)
    Rw = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], np.float64)
    for k in range(20):
        Tcl, Tlc, Tcw_last, Tcw_cur, vel = _scenario(rng, synthetic)
        ratio = float(rng.uniform(0.1, 1.2))
        st0 = pkg.pack_lidar_state(np.eye(3), np.zeros(3), np.eye(3), rng.normal(0, 0.1, 3))
        st, pos = oracle.update_lidar_pose(Tcw_last, vel, ratio, Tcl, st0)
        got_st, got_pos = pkg.capi.lidar_update_pose(Tcw_last, vel, ratio, Tcl, st0)
        assert np.array_equal(st, got_st) and np.array_equal(pos, got_pos)
        # float64: Twc = Tcw_last^-1 * expm(ratio * logm(velocity^-1)), Mwl = Twc * Tcl, state = Rw2_w1 * Mwl
        from scipy.linalg import expm, logm
        Twc = np.linalg.inv(mat(Tcw_last)) @ expm(ratio * np.real(logm(np.linalg.inv(mat(vel)))))
        Mwl = Twc @ mat(Tcl)
        assert np.abs(st[:9].reshape(3, 3) - Rw @ Mwl[:3, :3]).max() < 2e-5 and np.abs(st[9:12] - Rw @ Mwl[:3, 3]).max() < 2e-4
        assert np.allclose(pos, st[9:12] + st[:9].reshape(3, 3) @ st0[21:24], atol=1e-12)
        assert np.array_equal(st[12:], st0[12:])           # the offsets are not touched
        # InterpolateSE3 and the two chains
        t = float(rng.uniform(0, 1))
        assert np.array_equal(pkg.capi.se3_interpolate(Tcw_last, Tcw_cur, t), oracle.se3f_ops(Tcw_last, Tcw_cur, t)["interpolate"])
        for frame in (Tcw_cur, Tcw_last):
            want = oracle.sync_transform(frame, Tcw_last, Tcw_cur, ratio, Tlc, Tcl)
            assert np.array_equal(pkg.capi.lidar_sync_transform(frame, Tcw_last, Tcw_cur, ratio, Tlc, Tcl), want)
            Ai, Bi = np.linalg.inv(mat(Tcw_last)), np.linalg.inv(mat(Tcw_cur))
            Ri = Slerp([0, 1], Rotation.from_matrix(np.stack([Ai[:3, :3], Bi[:3, :3]])))([ratio] if ratio <= 1 else [1.0]).as_matrix()[0]
            if ratio <= 1:
                Mi = np.eye(4); Mi[:3, :3] = Ri; Mi[:3, 3] = Ai[:3, 3] + ratio * (Bi[:3, 3] - Ai[:3, 3])
                assert close(want, mat(Tlc) @ mat(frame) @ Mi @ mat(Tcl), 1e-4)
        rel, ref = rand_pose(rng, 0.05, 1.0), rand_pose(rng, 0.3, 20.0)
        want = oracle.keyframe_transform(Tcw_cur, rel, ref, Tlc, Tcl)
        assert np.array_equal(pkg.capi.lidar_keyframe_transform(Tcw_cur, rel, ref, Tlc, Tcl), want)
        assert close(want, mat(Tlc) @ mat(Tcw_cur) @ np.linalg.inv(mat(rel) @ mat(ref)) @ mat(Tcl), 1e-4)
    # a scan taken exactly at the current frame, paired with it: the cloud does not move (Tlc * Tcw * Twc * Tcl = identity)
    ident = oracle.sync_transform(Tcw_cur, Tcw_last, Tcw_cur, 1.0, Tlc, Tcl)
    assert close(ident, np.eye(4), 1e-4)


def test_oracle_transform_point_cloud(oracle):
    rng = np.random.default_rng(2)
    p = np.zeros(500, oracle.POINT_DTYPE)
    for f in ("x", "y", "z", "intensity", "curvature", "normal_x"):
        p[f] = rng.normal(0, 20, 500)
    T = rand_pose(rng)
    out = oracle.transform_point_cloud(p, T)
    xyz = np.stack([p["x"], p["y"], p["z"]], 1).astype(np.float64) @ mat(T)[:3, :3].T + mat(T)[:3, 3]
    assert np.abs(np.stack([out["x"], out["y"], out["z"]], 1) - xyz).max() < 1e-4
    assert np.array_equal(out["intensity"], p["intensity"]) and np.all(out["curvature"] == 0) and np.all(out["normal_x"] == 0) and np.all(out["pad0"] == 1)


@pytest.mark.gpu
def test_transform_point_cloud_gpu(pkg, oracle):
    rng = np.random.default_rng(3)
    for n in (1, 255, 256, 70001):
        p = np.zeros(n, oracle.POINT_DTYPE)
        for f in ("x", "y", "z", "intensity", "curvature", "normal_z"):
            p[f] = rng.normal(0, 30, n)
        T = rand_pose(rng)
        assert np.array_equal(pkg.capi.transform_point_cloud(p, T), oracle.transform_point_cloud(p, T))
    assert len(pkg.capi.transform_point_cloud(p[:0], T)) == 0


@pytest.mark.gpu
def test_transform_features_batch_gpu(pkg, oracle, synthetic):
    """SyncWithLidar's cloud for every scan of a batch: the front end's selections are read where they lie on the device."""
    import torch
    S = 3
    fe = pkg.LidarFrontEnd(max_points_per_scan=140000, max_scans=S)
    scene = synthetic.Scene(4)
    raws = [synthetic.lidar_scan(scene, f) for f in (1, 2, 3)]
    states = np.stack([pkg.pack_lidar_state(*synthetic.lidar_state(f)[:2]) for f in (1, 2, 3)])
    m = pkg.LidarMap(); m.Build(synthetic.lidar_map(scene, x_from=-50.0, x_to=80.0))
    raw = np.concatenate(raws)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in raws])]).astype(np.int32)
    dev = torch.from_numpy(raw.view(np.uint8)).cuda()
    counts, ori, _ = fe.frontend_batch(dev.data_ptr(), offs, [m] * S, states, want_points=True)
    rng = np.random.default_rng(5)
    T = np.stack([rand_pose(rng, 0.2, 3.0) for _ in range(S)])
    got = pkg.capi.lidar_transform_features_batch(fe, [2, 0], T[[2, 0]])
    for g, s in zip(got, (2, 0)):
        n = counts[2][s]
        assert len(g) == n and n > 1000
        assert np.array_equal(g, oracle.transform_point_cloud(ori[s, :n], T[s]))
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.lidar_transform_features_batch(fe, [5], T[:1])
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.lidar_transform_features_batch(fe, [0], T[:1], capacity=10)
