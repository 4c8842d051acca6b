"""k_pose_inertial alone: one frame (the single-sequence regime) and a batch of 512, ms per call and the kernel's own duration.
python tools/time_pose_inertial.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import tc2li_loader; pkg = tc2li_loader.load()
from tc2li_slam_amd import synthetic
def problem(seed, last_frame, n_points):
    w = synthetic.pose_inertial_problem(seed, last_frame=last_frame, n_points=n_points)
    p = pkg.capi.Preintegrated(w["bias6"], *synthetic.IMU_NOISE)
    p.preintegrate(w["samples"], w["t1"], w["t2"])
    w["pre"] = p; w["last_frame"] = last_frame; w["edges"] = pkg.pack_ba_edges(w["edges"])
    return w
for last in (True, False):
    ws = [problem(s, last, 1200) for s in range(4)]
    for n in (1, 512):
        items = [ws[k % 4] for k in range(n)]
        f = lambda: pkg.capi.pose_inertial_optimization_batch(items, ws[0]["calib24"], ws[0]["cam"])
        for _ in range(3): f()
        pkg.capi.profile_enable(True)
        for _ in range(5): f()
        pkg.capi.profile_enable(False)
        rep = pkg.capi.profile_report()
        for name, (cnt, ms) in rep.items():
            if "pose_inertial" in name:
                print("last_frame=%d, %3d frames: %-24s %.3f ms per launch" % (last, n, name, ms / cnt))
