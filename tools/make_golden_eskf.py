"""Golden vectors of the LiDAR-inertial ESKF update (esekf::update_iterated_dyn_share_modified with h_share_model):
tests/golden/eskf_a.npz = map points, one down-sampled body scan (every 3rd point), a perturbed state and covariance, and the oracle's
updated state / covariance / iteration bookkeeping with and without extrinsic estimation.  The reference ships no vectors for this path;
these are made here from the CPU oracle and committed with this script.
Run from the repository root:  python tools/make_golden_eskf.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tc2li_loader  # noqa: E402

pkg = tc2li_loader.load()
from tc2li_slam_amd import synthetic  # noqa: E402
from oracle import pyoracle  # noqa: E402
from test_eskf import scene_problem  # noqa: E402  (the seeded scene of the parity test)

pyoracle.build()
world0, down1, xt = scene_problem(pyoracle, synthetic, pkg, every=3)
rng = np.random.default_rng(41)
out = dict(map_points=world0, body=down1, x_true=xt)
for i, ext in enumerate((False, True)):
    d0 = np.concatenate([rng.normal(0, 0.08, 3), rng.normal(0, 0.01, 3), rng.normal(0, 0.002, 3) * ext, rng.normal(0, 0.01, 3) * ext,
                         rng.normal(0, 0.05, 3), np.zeros(8)])
    xe = pyoracle.eskf_boxplus(xt, d0)
    A = rng.normal(0, 1, (23, 23))
    P = A @ A.T * 1e-5 + np.diag([1e-2] * 3 + [1e-3] * 3 + [1e-5] * 6 + [1e-2] * 3 + [1e-4] * 6 + [1e-5] * 2)
    xs, Ps, info = pyoracle.eskf_update(xe, P, pyoracle.KdTree(world0), down1, max_iter=4, extrinsic_est_en=ext)
    out.update({"x_%d" % i: xe, "P_%d" % i: P, "ext_%d" % i: np.int32(ext), "out_x_%d" % i: xs, "out_P_%d" % i: np.asarray(Ps).reshape(23, 23),
                "out_info_%d" % i: np.array([info["calls"], info["searches"], info["converged"], int(info["finished"]), info["effct_feat_num"]], np.int64),
                "out_res_%d" % i: np.float64(info["res_mean_last"])})
    print("case", i, info)
path = os.path.join(ROOT, "tests", "golden", "eskf_a.npz")
np.savez_compressed(path, n_cases=np.int32(2), **out)
print("eskf_a", os.path.getsize(path) // 1024, "KiB")
