"""One lock-step group of N local LV-BA windows at the benched shape, alone on the GPU: ms per call and the kernels' own durations per
call (tc2li_profile_*), for the small-batch regime (64 sequences per GPU = groups of 5-16 windows).  python tools/time_ba_rounds.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import tc2li_loader; pkg = tc2li_loader.load()
from tc2li_slam_amd import synthetic
base = []
for seed in range(4):
    w = synthetic.ba_window(seed, n_opt=12, n_fix=20, n_points=3000, pose_noise=(0.1, 0.01))
    e = pkg.pack_ba_edges(w["edges"])
    last = len(w["poses"]) - 1
    win = list(range(last, last - 6, -1))
    base.append(dict(poses=w["poses"], fixed=w["fixed"], points=w["points"], edges=e, win_pose=win, clouds=synthetic.ba_window_clouds(w, win, n_points=3000), Tcl7=synthetic.TCL7, weight=1.0))
for N in [int(a) for a in sys.argv[1:]] or [5, 16, 43]:
    b = pkg.capi.BaBatch([base[k % 4] for k in range(N)], w["cam"])
    for _ in range(3): b.run_group(0)
    reps = 10
    t = time.perf_counter()
    for _ in range(reps): b.run_group(0)
    ms = (time.perf_counter() - t) * 1e3 / reps
    print("N = %d windows, one group: %.3f ms per call (%d iterations, %d trials in window 0)" % (N, ms, b.stats[0].iterations, b.stats[0].trials))
    pkg.capi.profile_enable(True)
    for _ in range(3): b.run_group(0)
    pkg.capi.profile_enable(False)
    rep = pkg.capi.profile_report()
    tot = sum(v[1] for v in rep.values()) / 3
    print("   kernel time per call %.3f ms in %d launches" % (tot, sum(v[0] for v in rep.values()) // 3))
    for name, (n, t_ms) in sorted(rep.items(), key=lambda kv: -kv[1][1])[:22]:
        print("   %-34s %4d launches per call, %7.1f us each, %7.3f ms per call" % (name, n // 3, 1e3 * t_ms / n, t_ms / 3))
