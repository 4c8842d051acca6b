"""Evaluation plumbing: camera trajectories in the two text formats the reference writes and the
absolute trajectory error between two of them.
  save_tum    System::SaveTrajectoryTUM   (SF/src/System.cc:379-432): "t tx ty tz qx qy qz qw" of Twc relative to the first pose,
              time relative to the first stamp, fixed notation, 6 / 9 decimals
  save_kitti  System::SaveTrajectoryKITTI (SF/src/System.cc:497-548): the 3 x 4 matrix [Rwc | twc] row-major, 9 decimals
  ate_rmse    root-mean-square translational error after the least-squares rigid alignment (Horn / Umeyama, no scale) -- the
              "ATE" of BASELINE.json's metric
Poses are 7-vectors (qx, qy, qz, qw, tx, ty, tz) of Tcw, as everywhere in this repository.
Usage: python tools/trajectory.py ate a.txt b.txt   (both in KITTI or both in TUM format)"""
import sys

import numpy as np


def quat_to_R(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def R_to_quat(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        return np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    i = int(np.argmax(np.diag(R)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
    q = np.zeros(4)
    q[i] = 0.25 * s
    q[3] = (R[k, j] - R[j, k]) / s
    q[j] = (R[j, i] + R[i, j]) / s
    q[k] = (R[k, i] + R[i, k]) / s
    return q


def inverse_poses(poses7):
    """Tcw (7-vectors) -> list of (Rwc, twc)."""
    out = []
    for p in np.asarray(poses7, np.float64).reshape(-1, 7):
        R = quat_to_R(p[:4])
        out.append((R.T, -R.T @ p[4:]))
    return out


def relative_to_first(poses7):
    """Twc of every frame in the frame of the first camera (Tcw * Two of the writers)."""
    tw = inverse_poses(poses7)
    R0, t0 = tw[0]
    return [(R0.T @ R, R0.T @ (t - t0)) for R, t in tw]


def save_kitti(path, poses7):
    with open(path, "w") as f:
        for R, t in relative_to_first(poses7):
            f.write(" ".join("%.9f" % v for v in np.concatenate([np.column_stack([R, t])]).ravel()) + "\n")


def save_tum(path, poses7, stamps):
    with open(path, "w") as f:
        for (R, t), ts in zip(relative_to_first(poses7), stamps):
            q = R_to_quat(R)
            f.write("%.6f %s\n" % (ts - stamps[0], " ".join("%.9f" % v for v in np.concatenate([t, q]))))


def load_positions(path):
    rows = np.loadtxt(path, ndmin=2)
    if rows.shape[1] == 12:
        return rows[:, [3, 7, 11]]
    if rows.shape[1] == 8:
        return rows[:, 1:4]
    raise ValueError("%s: neither KITTI (12 columns) nor TUM (8 columns)" % path)


def align_rigid(a, b):
    """R, t minimising sum |R a_i + t - b_i|^2 (Horn's closed form through the SVD)."""
    ca, cb = a.mean(0), b.mean(0)
    H = (a - ca).T @ (b - cb)
    U, _, Vt = np.linalg.svd(H)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    return R, cb - R @ ca


def ate_rmse(est_xyz, ref_xyz, align=True):
    a, b = np.asarray(est_xyz, np.float64), np.asarray(ref_xyz, np.float64)
    if align and len(a) >= 3:
        R, t = align_rigid(a, b)
        a = a @ R.T + t
    return float(np.sqrt(np.mean(np.sum((a - b) ** 2, 1))))


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "ate":
        print("ATE rmse %.6f m" % ate_rmse(load_positions(sys.argv[2]), load_positions(sys.argv[3])))
    else:
        print(__doc__)
