"""Dumps the results of the two pose optimisation kernels on fixed synthetic problems (bit-level comparisons between two builds):
python tools/dbg/pose_bits.py out.npz"""
import sys
import numpy as np
sys.path.insert(0, ".")
import tc2li_loader
pkg = tc2li_loader.load()
from importlib import import_module
from tc2li_slam_amd import synthetic
sys.path.insert(0, "tests")
w = synthetic.ba_window(4, n_opt=8, n_fix=4, n_points=1200, outlier_frac=0.1)
ks = list(range(4, 12))


def frame_problem(w, k):
    e = w["edges"]
    m = e[:, 1].astype(int) == k
    ed = e[m].copy()
    Xw = w["points"][ed[:, 0].astype(int)]
    ed[:, 0] = np.arange(len(ed)); ed[:, 1] = 0
    return Xw, ed


probs = [frame_problem(w, k) for k in ks]
offs = np.concatenate([[0], np.cumsum([len(e) for _, e in probs])])
Xw = np.concatenate([x for x, _ in probs]); ed = np.concatenate([e for _, e in probs])
poses, out, inl = pkg.pose_optimization_batch(w["poses"][ks], offs, Xw, pkg.pack_ba_edges(ed), w["cam"])
res = dict(poses=poses, out=out, inl=inl)
items = []
for seed, last in [(0, False), (1, True), (2, True)]:
    q = synthetic.pose_inertial_problem(seed, last_frame=last, n_points=700)
    p = pkg.capi.Preintegrated(q["bias6"], *synthetic.IMU_NOISE)
    p.preintegrate(q["samples"], q["t1"], q["t2"])
    q["pre"] = p; q["last_frame"] = last; q["edges"] = pkg.pack_ba_edges(q["edges"])
    items.append(q)
got = pkg.capi.pose_inertial_optimization_batch(items, items[0]["calib24"], items[0]["cam"])
for i, g in enumerate(got):
    res["pi%d_cur" % i] = g[0]; res["pi%d_oth" % i] = g[1]; res["pi%d_out" % i] = g[2]; res["pi%d_prior" % i] = g[3]
np.savez(sys.argv[1], **res)
print("saved", sys.argv[1], {k: np.asarray(v).shape for k, v in res.items()})
