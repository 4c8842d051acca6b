"""Times the HIP local BA / pose optimisation against the CPU oracle on the BASELINE-sized synthetic window."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tc2li_loader
pkg = tc2li_loader.load()
from tc2li_slam_amd import synthetic
from oracle import pyoracle as O

w = synthetic.ba_window(0, n_opt=12, n_fix=20, n_points=3000)
e = pkg.pack_ba_edges(w["edges"])
print("window: poses", len(w["poses"]), "points", len(w["points"]), "edges", len(e))
for _ in range(2):
    pkg.local_bundle_adjustment(w["poses"], w["fixed"], w["points"], e, w["cam"])
t = time.perf_counter(); n = 10
for _ in range(n):
    r = pkg.local_bundle_adjustment(w["poses"], w["fixed"], w["points"], e, w["cam"])
tg = (time.perf_counter() - t) / n
t = time.perf_counter()
for _ in range(5):
    o = O.local_ba(w["poses"], w["fixed"], w["points"], w["edges"], w["cam"])
tc = (time.perf_counter() - t) / 5
print("local BA: gpu %.2f ms (iters %d trials %d)  cpu oracle %.2f ms  speedup %.1fx" % (1e3 * tg, r[4].iterations, r[4].trials, 1e3 * tc, tc / tg))
# pose optimisation, batch of 32 frames
k = len(w["poses"]) - 1
ed = w["edges"][w["edges"][:, 1] == k].copy(); Xw = w["points_true"][ed[:, 0].astype(int)]; ed[:, 0] = np.arange(len(ed)); ed[:, 1] = 0
F = 32
offs = np.arange(F + 1) * len(ed); XW = np.tile(Xw, (F, 1)); ED = pkg.pack_ba_edges(np.tile(ed, (F, 1))); P0 = np.tile(w["poses"][k], (F, 1))
for _ in range(2):
    pkg.pose_optimization_batch(P0, offs, XW, ED, w["cam"])
t = time.perf_counter()
for _ in range(10):
    pkg.pose_optimization_batch(P0, offs, XW, ED, w["cam"])
tg = (time.perf_counter() - t) / 10
t = time.perf_counter()
for _ in range(F):
    O.pose_optimization(w["poses"][k], Xw, ed, w["cam"])
tc = time.perf_counter() - t
print("pose optimisation x%d (%d edges each): gpu %.2f ms  cpu oracle %.2f ms  speedup %.1fx" % (F, len(ed), 1e3 * tg, 1e3 * tc, tc / tg))
# local BA with the LiDAR edge (LocalLVBundleAdjustment): 6 window keyframes, dense surface clouds
w = synthetic.ba_window(0, n_opt=12, n_fix=20, n_points=3000, pose_noise=(0.1, 0.01))
e = pkg.pack_ba_edges(w["edges"])
last = len(w["poses"]) - 1
win = list(range(last, last - 6, -1))
for npts in (3000, 20000):
    clouds = synthetic.ba_window_clouds(w, win, n_points=npts)
    for _ in range(2):
        r = pkg.capi.local_lv_bundle_adjustment(w["poses"], w["fixed"], w["points"], e, w["cam"], win, clouds, synthetic.TCL7, 1.0)
    t = time.perf_counter()
    for _ in range(10):
        r = pkg.capi.local_lv_bundle_adjustment(w["poses"], w["fixed"], w["points"], e, w["cam"], win, clouds, synthetic.TCL7, 1.0)
    tg = (time.perf_counter() - t) / 10
    t = time.perf_counter()
    pkg.capi.lidar_planes_host(w["poses"], win, clouds, synthetic.TCL7)
    tp = time.perf_counter() - t
    t = time.perf_counter()
    for _ in range(3):
        o = O.local_ba_lidar(w["poses"], w["fixed"], w["points"], w["edges"], w["cam"], win, clouds, synthetic.TCL7, 1.0)
    tc = (time.perf_counter() - t) / 3
    print("local LV-BA, %d pts/cloud, %d planes: gpu %.2f ms (host plane extraction %.2f ms; iters %d trials %d, %d Hessians)  cpu oracle %.2f ms  speedup %.1fx"
          % (npts, r[5].n_planes, 1e3 * tg, 1e3 * tp, r[4].iterations, r[4].trials, r[5].hessian_evaluations, 1e3 * tc, tc / tg))
