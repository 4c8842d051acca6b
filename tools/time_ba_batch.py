import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import tc2li_loader; pkg = tc2li_loader.load()
from tc2li_slam_amd import synthetic
wins=[]
for seed in range(8):
    w = synthetic.ba_window(seed, n_opt=12, n_fix=20, n_points=3000, pose_noise=(0.1, 0.01))
    e = pkg.pack_ba_edges(w["edges"])
    last = len(w["poses"]) - 1
    win = list(range(last, last - 6, -1))
    wins.append(dict(poses=w["poses"], fixed=w["fixed"], points=w["points"], edges=e, win_pose=win, clouds=synthetic.ba_window_clouds(w, win, n_points=3000), Tcl7=synthetic.TCL7, weight=1.0))
b = pkg.capi.BaBatch(wins, w["cam"])
for _ in range(3): b.run(max_concurrency=16)
t=time.perf_counter()
for _ in range(10): b.run(max_concurrency=16)
print("batch of 8: %.2f ms" % ((time.perf_counter()-t)*100))
