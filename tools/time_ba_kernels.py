"""Per-kernel device time of ONE lock-step group of LV-BA windows with the GPU to itself (tc2li_profile_*: start / stop events of every
dispatch).  python tools/time_ba_kernels.py [n_windows] -- run with TC2LI_BA_LOCKSTEP_GROUPS=1 (one group, one stream)."""
import os, sys
os.environ.setdefault("TC2LI_BA_LOCKSTEP_GROUPS", "1")
sys.path.insert(0, os.getcwd())
import tc2li_loader; pkg = tc2li_loader.load()
from tc2li_slam_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 43
base = []
for seed in range(4):
    w = synthetic.ba_window(seed, n_opt=12, n_fix=20, n_points=3000, pose_noise=(0.1, 0.01))
    last = len(w["poses"]) - 1
    win = list(range(last, last - 6, -1))
    base.append(dict(poses=w["poses"], fixed=w["fixed"], points=w["points"], edges=pkg.pack_ba_edges(w["edges"]), win_pose=win,
                     clouds=synthetic.ba_window_clouds(w, win, n_points=3000), Tcl7=synthetic.TCL7, weight=1.0))
b = pkg.capi.BaBatch([base[k % 4] for k in range(n)], w["cam"])
for _ in range(2):
    b.run(max_concurrency=16)
pkg.capi.profile_enable(True)
reps = 3
for _ in range(reps):
    b.run(max_concurrency=16)
pkg.capi.profile_enable(False)
rep = pkg.capi.profile_report()
tot = sum(ms for _, ms in rep.values())
print("%d windows, one lock-step group: %.3f ms of kernel time per batch; iterations/trials of window 0: %d/%d" % (n, tot / reps, b.stats[0].iterations, b.stats[0].trials))
for name, (calls, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1])[:18]:
    print("%-44s %5d launches  %8.1f us avg  %6.2f %%" % (name.split("(")[0][:44], calls, 1e3 * ms / calls, 100 * ms / tot))
