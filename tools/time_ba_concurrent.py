"""G lock-step groups of N windows each, run concurrently from G host threads (the mapping workers of the bench), alone on the GPU:
ms per call of a group against the one-group figure.  python tools/time_ba_concurrent.py N G [G ...]"""
import os, sys, time, threading
sys.path.insert(0, os.getcwd())
import tc2li_loader; pkg = tc2li_loader.load()
from tc2li_slam_amd import synthetic
base = []
for seed in range(4):
    w = synthetic.ba_window(seed, n_opt=12, n_fix=20, n_points=3000, pose_noise=(0.1, 0.01))
    e = pkg.pack_ba_edges(w["edges"])
    last = len(w["poses"]) - 1
    win = list(range(last, last - 6, -1))
    base.append(dict(poses=w["poses"], fixed=w["fixed"], points=w["points"], edges=e, win_pose=win, clouds=synthetic.ba_window_clouds(w, win, n_points=3000), Tcl7=synthetic.TCL7, weight=1.0))
N = int(sys.argv[1])
for G in [int(a) for a in sys.argv[2:]]:
    batches = [pkg.capi.BaBatch([base[k % 4] for k in range(N)], w["cam"]) for _ in range(G)]
    reps = 12
    def work(g):
        for _ in range(3): batches[g].run_group(g)
    ts = [threading.Thread(target=work, args=(g,)) for g in range(G)]
    [t.start() for t in ts]; [t.join() for t in ts]
    out = [0.0] * G
    def timed(g):
        t = time.perf_counter()
        for _ in range(reps): batches[g].run_group(g)
        out[g] = (time.perf_counter() - t) * 1e3 / reps
    t0 = time.perf_counter()
    ts = [threading.Thread(target=timed, args=(g,)) for g in range(G)]
    [t.start() for t in ts]; [t.join() for t in ts]
    wall = (time.perf_counter() - t0) * 1e3 / reps
    print("N = %d windows x G = %d concurrent groups: %.3f ms per call (slowest group), %.1f windows per ms" % (N, G, max(out), N * G / wall))
