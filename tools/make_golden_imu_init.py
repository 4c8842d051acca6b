"""Golden vectors of the IMU initialisation: tests/golden/imu_init_a.npz = one problem (14 keyframes of the
synthetic drive in a tilted visual world, pre-integrations at zero bias packed as 298 floats) with the oracle's first gravity estimate
(LocalMapping::InitializeIMU) and the oracle's Optimizer::InertialOptimization result for the reference's priors and for mild ones.
The reference ships no vectors for this path; these are made here from the CPU oracle and committed with this script.
Run from the repository root:  python tools/make_golden_imu_init.py"""
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
w = synthetic.imu_init_problem(40, n_kf=14)
n = len(w["Rwb"])
pre = [np.zeros(298, np.float32)]
for s, t1, t2 in w["samples"]:
    _, f = pyoracle.imu_preintegrate(s, t1, t2, np.zeros(6), *synthetic.IMU_NOISE)
    pre.append(pyoracle.pack_preintegrated(f, np.zeros(6)))
pre = np.stack(pre)
kf = np.zeros((n, 33))
kf[:, 12:21], kf[:, 21:24] = w["Rwb"].reshape(n, 9), w["twb"]
vel0, Rwg0 = pyoracle.initial_gravity_direction(kf, pre)
kf[:, 24:27] = vel0
out = dict(Rwb=w["Rwb"], twb=w["twb"], pre298=pre, vel0=vel0, Rwg0=Rwg0, Rwg_true=w["Rwg_true"], bg_true=w["bg_true"])
for tag, (pg, pa) in (("ref", (1e2, 1e6)), ("mild", (1.0, 1e3))):
    o = pyoracle.inertial_optimization(kf, pre, Rwg0, 1.0, np.zeros(3), np.zeros(3), priorG=pg, priorA=pa)
    out.update({"vel_" + tag: o[0][:, 24:27], "Rwg_" + tag: o[1], "bg_" + tag: o[3], "ba_" + tag: o[4], "counts_" + tag: np.array([o[5], o[6]], np.int64),
                "err_" + tag: np.array(o[7]), "priors_" + tag: np.array([pg, pa])})
    print(tag, "iterations", o[5], "trials", o[6], "chi2", o[7])
path = os.path.join(ROOT, "tests", "golden", "imu_init_a.npz")
np.savez_compressed(path, **out)
print("imu_init_a", os.path.getsize(path) // 1024, "KiB")
