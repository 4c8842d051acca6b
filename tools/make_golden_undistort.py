"""Golden vectors of the scan motion compensation (ImuProcess forward propagation + UndistortPcl):
tests/golden/undistort_a.npz = every 8th point of a preprocessed synthetic scan, the IMU samples of the sweep, the initial state, and the
oracle's propagated state, IMU poses and compensated points (in std::sort's order of the time offsets).  The reference ships no vectors
for this path; these are made here from the CPU oracle and committed with this script.
Run from the repository root:  python tools/make_golden_undistort.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tc2li_loader  # noqa: E402

tc2li_loader.load()
from tc2li_slam_amd import synthetic  # noqa: E402
from oracle import pyoracle  # noqa: E402
from test_undistort_gpu import imu_stream, initial_state, lidar_state24  # noqa: E402  (the seeded generators of the parity test)

pyoracle.build()
seed = 5
pts = pyoracle.lidar_preprocess(synthetic.lidar_scan(synthetic.Scene(seed), seed + 1))[::8].copy()
beg, end = 20.0, 20.1
imu = imu_stream(beg - 0.012, end + 0.004, seed=seed)
st0 = initial_state(seed)
last6 = np.array([0.1, -0.05, 0.02, 0.01, 0.0, 0.24])
st, poses = pyoracle.imu_propagate(st0, imu, beg, end, beg - 0.001, 9.81 / 9.79, last6)
out = pyoracle.undistort(pts, poses, lidar_state24(st))
print("points", len(pts), "imu poses", len(poses))
path = os.path.join(ROOT, "tests", "golden", "undistort_a.npz")
np.savez_compressed(path, points=pts, imu=imu, state0=st0, last6=last6, t=np.array([beg, end, beg - 0.001, 9.81 / 9.79]), out_state=st, out_poses=poses,
                    out_points=out)
print("undistort_a", os.path.getsize(path) // 1024, "KiB")
