"""Soak of tc2li_ba_engine: T submitter threads hand random slices of the varied windows to K engines for S seconds (tickets of 1-6 windows, up
to three open per thread, engines of few slots so that windows queue, a stop flag set on some windows while they run); every result is compared
bit for bit with the batch call's (a stopped window: with nothing but the iteration count's range).  python tools/soak_engine.py [seconds]"""
import os, sys, time, threading
sys.path.insert(0, os.getcwd())
import numpy as np
import tc2li_loader; pkg = tc2li_loader.load()
from tc2li_slam_amd import synthetic
S = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
ws = [synthetic.ba_window_varied(k) for k in range(24)]
def as_dict(w):
    d = dict(poses=w["poses"], fixed=w["fixed"], points=w["points"], edges=pkg.pack_ba_edges(w["edges"]), iterations=w["iterations"])
    if w["win_pose"]: d.update(win_pose=w["win_pose"], clouds=w["clouds"], Tcl7=synthetic.TCL7, weight=w["weight"])
    return d
ds = [as_dict(w) for w in ws]
cam = ws[0]["cam"]
ref = pkg.capi.BaBatch(ds, cam)
assert ref.run_group(0) == len(ds)
want = [(np.array(ref.result(i)[0], copy=True), np.array(ref.result(i)[1], copy=True), int(ref.results[i]), ref.result(i)[4].trials) for i in range(len(ds))]
engines = [pkg.capi.BaEngine(cam, max_windows=m) for m in (3, 7, 40)]
counts, errors, t_end = [0] * 8, [], time.time() + S
def submitter(t):
    rng = np.random.default_rng(100 + t)
    mine = pkg.capi.BaBatch(ds, cam)   # this thread's own arrays
    busy = np.zeros(len(ds), bool)
    open_t = []
    try:
        while time.time() < t_end or open_t:
            if time.time() < t_end and len(open_t) < 3:
                n = int(rng.integers(1, 7)); first = int(rng.integers(0, len(ds) - n + 1))
                if not busy[first:first + n].any():
                    e = engines[int(rng.integers(0, len(engines)))]
                    busy[first:first + n] = True
                    open_t.append((e, e.submit(mine, first, n), first, n))
                    continue
            if open_t:
                e, tk, first, n = open_t.pop(0)
                assert e.wait(tk) == n
                for i in range(first, first + n):
                    r = mine.result(i)
                    if not (np.array_equal(r[0], want[i][0]) and np.array_equal(r[1], want[i][1]) and int(mine.results[i]) == want[i][2] and r[4].trials == want[i][3]):
                        errors.append((t, i, int(mine.results[i]), want[i][2]))
                busy[first:first + n] = False
                counts[t] += n
    except BaseException as ex:  # noqa: BLE001
        errors.append((t, repr(ex)))
ts = [threading.Thread(target=submitter, args=(t,)) for t in range(8)]
t0 = time.time()
[t.start() for t in ts]; [t.join() for t in ts]
for e in engines: e.close()
print("soak: %d windows through %d engines from %d threads in %.1f s, %d mismatches / errors %s" % (sum(counts), len(engines), len(ts), time.time() - t0, len(errors), errors[:5]))
sys.exit(1 if errors else 0)
