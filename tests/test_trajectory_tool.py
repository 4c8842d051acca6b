"""tools/trajectory.py (evaluation plumbing): the two trajectory formats and the ATE."""
import os
import sys

import numpy as np
from scipy.spatial.transform import Rotation

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import trajectory as T  # noqa: E402


def random_poses(n, seed=0):
    rng = np.random.default_rng(seed)
    q = Rotation.from_rotvec(rng.normal(0, 0.3, (n, 3))).as_quat()
    return np.concatenate([q, np.cumsum(rng.normal(0, 0.5, (n, 3)), 0)], 1)


def test_writers_round_trip(tmp_path):
    poses = random_poses(20)
    T.save_kitti(str(tmp_path / "k.txt"), poses)
    T.save_tum(str(tmp_path / "t.txt"), poses, 100.0 + 0.1 * np.arange(20))
    k, t = T.load_positions(str(tmp_path / "k.txt")), T.load_positions(str(tmp_path / "t.txt"))
    assert np.allclose(k, t, atol=1e-8)
    assert np.allclose(k[0], 0)  # relative to the first camera, as the reference writes them
    # camera centres: twc = -R^T t, expressed in the first camera's frame
    R0 = T.quat_to_R(poses[0, :4])
    c = np.stack([-T.quat_to_R(p[:4]).T @ p[4:] for p in poses])
    assert np.allclose(k, (c - c[0]) @ R0.T, atol=1e-8)
    rows = np.loadtxt(str(tmp_path / "k.txt"))
    R1 = rows[5].reshape(3, 4)[:, :3]
    assert np.allclose(R1 @ R1.T, np.eye(3), atol=1e-8)
    tum = np.loadtxt(str(tmp_path / "t.txt"))
    assert np.allclose(tum[:, 0], 0.1 * np.arange(20), atol=1e-6) and np.allclose(np.linalg.norm(tum[:, 4:], axis=1), 1, atol=1e-8)
    # quaternion <-> matrix helpers
    for p in poses:
        assert np.allclose(T.quat_to_R(T.R_to_quat(T.quat_to_R(p[:4]))), T.quat_to_R(p[:4]), atol=1e-12)


def test_ate():
    rng = np.random.default_rng(1)
    a = np.cumsum(rng.normal(0, 1, (50, 3)), 0)
    R = Rotation.from_rotvec([0.3, -0.2, 0.9]).as_matrix()
    b = a @ R.T + np.array([5.0, -2.0, 1.0])
    assert T.ate_rmse(a, b) < 1e-12                      # a rigid motion is aligned away
    assert T.ate_rmse(a, b, align=False) > 1.0
    noise = rng.normal(0, 0.05, a.shape)
    e = T.ate_rmse(a + noise, b)
    assert 0.05 * np.sqrt(3) * 0.8 < e < 0.05 * np.sqrt(3) * 1.1
    assert T.ate_rmse(a[:2], a[:2] + 1.0) > 0            # fewer than three poses: no alignment
