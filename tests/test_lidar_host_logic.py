"""Host-only pieces of the LiDAR path (no GPU): lasermap_fov_segment and the IMU forward propagation against the oracle."""
import numpy as np


def test_fov_segment_matches_the_oracle(pkg, oracle):
    rng = np.random.default_rng(0)
    lm = pkg.capi.LocalMapBox()
    lm7 = np.zeros(7, np.float32)
    pos = np.zeros(3)
    moved = 0
    for _ in range(200):
        pos = pos + rng.normal([4.0, 0.5, 0.0], [2.0, 2.0, 0.3])
        got = pkg.capi.lidar_fov_segment(lm, pos, cube_len=200.0, det_range=100.0 / 3)
        lm7, want = oracle.fov_segment(lm7, pos, 200.0, 100.0 / 3)
        assert np.array_equal(got, want)
        assert np.array_equal(np.array(lm.vertex_min), lm7[:3]) and np.array_equal(np.array(lm.vertex_max), lm7[3:6])
        moved += len(got) > 0
    assert moved >= 3  # the cube was shifted several times along the way


def test_fov_segment_batch_equals_the_per_sequence_calls(pkg, oracle):
    """tc2li_lidar_fov_segment_batch: the cubes of n sequences in one call -- every sequence's boxes and cube are those of its own
    tc2li_lidar_fov_segment calls (and so the oracle's)."""
    F = 17
    rng = np.random.default_rng(3)
    cubes = (pkg.capi.LocalMapBox * F)()
    singles = [pkg.capi.LocalMapBox() for _ in range(F)]
    pos = np.zeros((F, 3))
    total = 0
    for _ in range(60):
        pos = pos + rng.normal([6.0, 1.0, 0.0], [3.0, 3.0, 0.3], (F, 3))
        boxes, counts = pkg.capi.lidar_fov_segment_batch(cubes, pos, cube_len=200.0, det_range=100.0 / 3)
        for s in range(F):
            want = pkg.capi.lidar_fov_segment(singles[s], pos[s], cube_len=200.0, det_range=100.0 / 3)
            assert counts[s] == len(want) and np.array_equal(boxes[s, :counts[s]], want)
            assert np.array_equal(np.array(cubes[s].vertex_min), np.array(singles[s].vertex_min))
            assert np.array_equal(np.array(cubes[s].vertex_max), np.array(singles[s].vertex_max))
        total += int(counts.sum())
    assert total >= F  # every cube was shifted along the way
    assert pkg.capi.lidar_fov_segment_batch((pkg.capi.LocalMapBox * 0)(), np.zeros((0, 3)))[1].shape == (0,)


def test_imu_forward_propagation_matches_the_oracle(pkg, oracle):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(1)
    for seed in range(3):
        t = np.arange(995, 1012) / 100.0
        imu = np.zeros((len(t), 7)); imu[:, 0] = t
        imu[:, 1:4] = [0.1, -0.2, 9.8] + rng.normal(0, 0.05, (len(t), 3))
        imu[:, 4:7] = [0.01, 0.02, 0.3] + rng.normal(0, 0.01, (len(t), 3))
        st = np.concatenate([rng.normal(0, 3, 3), Rotation.from_rotvec(rng.normal(0, 0.3, 3)).as_matrix().reshape(-1), [8, 0.2, 0], [1e-3, 0, -1e-3],
                             [0.01, 0.02, 0], [0, 0, -9.81], np.eye(3).reshape(-1), [0.1, 0, 0.05]])
        last6 = rng.normal(0, 0.1, 6)
        ws, wp = oracle.imu_propagate(st, imu, 10.0, 10.1, 9.999, 1.002, last6)
        gs, gp, glast = pkg.capi.lidar_imu_propagate(st, imu, 10.0, 10.1, 9.999, 1.002, last6)
        assert len(gp) == len(wp) >= 10
        assert np.allclose(gp, wp, rtol=1e-12, atol=1e-12) and np.allclose(gs, ws, rtol=1e-12, atol=1e-12)
        assert np.allclose(gp[-1][1:7], glast)  # acc_s_last / angvel_last carried to the next scan
        R = gs[3:12].reshape(3, 3)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12)
