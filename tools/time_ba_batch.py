"""The varied local-BA windows of the bench, alone on the GPU: ms per tc2li_local_bundle_adjustment_batch call of N windows (its three
lock-step groups), and G such calls side by side from G host threads on groups of their own (run_group: one group per call).
python tools/time_ba_batch.py"""
import os, sys, time, threading
sys.path.insert(0, os.getcwd())
import numpy as np
import tc2li_loader; pkg = tc2li_loader.load()
from tc2li_slam_amd import synthetic
ws = [synthetic.ba_window_varied(k) for k in range(64)]
def as_dict(w):
    d = dict(poses=w["poses"], fixed=w["fixed"], points=w["points"], edges=pkg.pack_ba_edges(w["edges"]), iterations=w["iterations"])
    if w["win_pose"]: d.update(win_pose=w["win_pose"], clouds=w["clouds"], Tcl7=synthetic.TCL7, weight=w["weight"])
    return d
ds = [as_dict(w) for w in ws]
cam = ws[0]["cam"]
for n in (16, 32, 64, 128):
    b = pkg.capi.BaBatch([ds[k % 64] for k in range(n)], cam)
    for _ in range(2): b.run(8)
    reps = 6
    t = time.perf_counter()
    for _ in range(reps): b.run(8)
    ms = (time.perf_counter() - t) * 1e3 / reps
    print("batch call of %3d windows: %7.3f ms  (%.3f ms per 16 windows)" % (n, ms, ms * 16 / n))
for n, G in ((16, 2), (16, 3), (16, 4), (8, 4), (6, 6), (8, 8)):
    batches = [pkg.capi.BaBatch([ds[(g * n + k) % 64] for k in range(n)], cam) for g in range(G)]
    def work(g, reps, out):
        t = time.perf_counter()
        for _ in range(reps): batches[g].run_group(g)
        out[g] = (time.perf_counter() - t) * 1e3 / reps
    out = [0.0] * G
    ts = [threading.Thread(target=work, args=(g, 2, out)) for g in range(G)]
    [t.start() for t in ts]; [t.join() for t in ts]
    t0 = time.perf_counter()
    ts = [threading.Thread(target=work, args=(g, 6, out)) for g in range(G)]
    [t.start() for t in ts]; [t.join() for t in ts]
    wall = (time.perf_counter() - t0) * 1e3 / 6
    print("%d groups of %2d windows side by side: %7.3f ms per round of calls (%.3f ms per 16 windows), slowest group %.3f ms" % (G, n, wall, wall * 16 / (n * G), max(out)))
