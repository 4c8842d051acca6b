"""Golden vectors of the tracking data path (SURVEY 8a rows a9 + a10 composed: Tracking::TrackWithMotionModel = SearchByProjection
against the last frame, with the wide-window retry, then Optimizer::PoseOptimization): tests/golden/tracking_a.npz = a stereo crop, the
last frame's points / keypoints / descriptors, the predicted pose, and the oracle's matches, inlier count and optimised pose.  The
reference ships no vectors for this path; these are made here from the CPU oracle and committed with this script.
Run from the repository root:  python tools/make_golden_tracking.py"""
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
left, right = synthetic.stereo_pair(33)
x0, y0, w, h = 250, 60, 560, 240
L, R = np.ascontiguousarray(left[y0:y0 + h, x0:x0 + w]), np.ascontiguousarray(right[y0:y0 + h, x0:x0 + w])
N = 800
ol, orr = pyoracle.OrbOracle(nfeatures=N), pyoracle.OrbOracle(nfeatures=N)
_, kl, dl = ol.extract(L)
_, kr, dr = orr.extract(R)
bf = float(np.float32(synthetic.BF))
b = float(np.float32(synthetic.BF) / np.float32(synthetic.FX))
u_right, depth, _ = pyoracle.stereo_match(ol, orr, kl, dl, kr, dr, bf, b)
fx, fy = np.float32(synthetic.FX), np.float32(synthetic.FY)
cx, cy = np.float32(synthetic.CX - x0), np.float32(synthetic.CY - y0)  # principal point of the crop
rng = np.random.default_rng(330)
n = len(kl)
order = rng.permutation(n)
lk = kl[order].copy()
lk["angle"] = (lk["angle"] + rng.normal(0, 3, n).astype(np.float32)) % np.float32(360)
z = depth[order]
has_point = (z > 0).astype(np.uint8)
zz = np.where(z > 0, z, 1).astype(np.float32)
Xw = np.stack([(lk["x"] - cx) * zz / fx, (lk["y"] - cy) * zz / fy, zz], 1).astype(np.float32)
outlier = (rng.random(n) < 0.05).astype(np.uint8)
md = dl[order].copy()
for i in range(n):
    for bit in rng.choice(256, size=int(rng.integers(0, 20)), replace=False):
        md[i, bit // 8] ^= np.uint8(1 << (bit % 8))
last_pose = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
ang = 0.001
pred = np.array([0, np.sin(ang / 2), 0, np.cos(ang / 2), 0.02, -0.01, -0.05], np.float32)
cam5 = np.float32([fx, fy, cx, cy, bf]).astype(np.float64)
scales, _, inv_sigma2 = ol.tables()[0], None, None
s, per_level, extra = ol.tables()
sigma2 = (s * s).astype(np.float32)
inv_sigma2 = (np.float32(1) / sigma2).astype(np.float32)
pose, matches, n_matches, n_inliers = pyoracle.track_motion_model(kl, dl, u_right, w, h, s, inv_sigma2, pred, last_pose, cam5, b, 7.0, has_point, outlier,
                                                                  Xw, lk, md)[:4]
print("keypoints", n, "matches", n_matches, "inliers", n_inliers, "pose", np.round(pose, 5))
kpf = lambda k: np.stack([k[f].astype(np.float32) for f in ("x", "y", "size", "angle", "response")] + [k["octave"].astype(np.float32)], 1)
path = os.path.join(ROOT, "tests", "golden", "tracking_a.npz")
np.savez_compressed(path, left=L, right=R, nfeatures=np.int32(N), bf=np.float32(bf), b=np.float32(b), cam5=cam5, th=np.float32(7.0),
                    last_keys=kpf(lk), last_desc=md, last_has_point=has_point, last_outlier=outlier, last_Xw=Xw, last_pose7=last_pose, pred7=pred,
                    inv_sigma2=inv_sigma2, scales=s, out_pose7=np.asarray(pose, np.float64), out_matches=np.asarray(matches, np.int32),
                    out_n_matches=np.int32(n_matches), out_n_inliers=np.int32(n_inliers))
print("tracking_a", os.path.getsize(path) // 1024, "KiB")
