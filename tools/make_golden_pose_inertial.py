"""Golden vectors of Optimizer::PoseInertialOptimizationLastKeyFrame / LastFrame (row a10'): tests/golden/pose_inertial_a.npz = two
problems (keyframe form, previous-frame form) with the oracle's pre-integration packed as 298 floats, and the oracle's optimised states,
outlier flags, counts and new prior.  The reference ships no vectors for this path; these are made here from the CPU oracle and committed
with this script.  Run from the repository root:  python tools/make_golden_pose_inertial.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tc2li_loader  # noqa: E402

tc2li_loader.load()
from tc2li_slam_amd import synthetic  # noqa: E402
from oracle import pyoracle  # noqa: E402

pyoracle.build()
out = {}
for i, last in enumerate((False, True)):
    w = synthetic.pose_inertial_problem(20 + i, n_points=350, last_frame=last)
    _, f = pyoracle.imu_preintegrate(w["samples"], w["t1"], w["t2"], w["bias6"], *synthetic.IMU_NOISE)
    pre298 = pyoracle.pack_preintegrated(f, w["bias6"])
    cur, oth, outlier, prior, rv, counts = pyoracle.pose_inertial(w["cur33"], w["other33"], last, w["prior246"], w["calib24"], pre298, pre298, w["Xw"], w["edges"],
                                                                w["close"], w["cam"])
    out.update({"cur33_%d" % i: w["cur33"], "other33_%d" % i: w["other33"], "prior246_%d" % i: w["prior246"] if last else np.zeros(0), "calib24": w["calib24"],
                "pre298_%d" % i: pre298, "Xw_%d" % i: w["Xw"], "edges_%d" % i: w["edges"], "close_%d" % i: w["close"], "cam": w["cam"],
                "out_cur_%d" % i: cur, "out_other_%d" % i: oth, "out_outlier_%d" % i: outlier, "out_prior_%d" % i: prior,
                "out_counts_%d" % i: np.array([rv, *counts], np.int64)})
    print("case", i, "returns", rv, counts, "outliers", int(outlier.sum()))
path = os.path.join(ROOT, "tests", "golden", "pose_inertial_a.npz")
np.savez_compressed(path, n_cases=np.int32(2), **out)
print("pose_inertial_a", os.path.getsize(path) // 1024, "KiB")
