"""Generates the golden vectors of SURVEY.md section 8c items (2)-(5) under tests/golden/ from the CPU oracle on seeded
synthetic inputs (the reference ships none, so they are created here and committed together with this script):
  stereo_a.npz    stereo pair crop -> keypoints / descriptors of both images, uRight / depth / SAD per left keypoint
  lidar_a.npz     map points + one down-sampled body scan + state -> selection mask, plane (n, d), pd2, world points
  ba_a.npz        small local window -> optimised poses / points, per-iteration chi2 / lambda / trial trace (visual only and
                  with the LiDAR edge over 4 window keyframes)
  balm_a.npz      4 keyframes looking at two planes -> residual, JacT, Hessian of the BALM edge
  inertial_a.npz  visual-inertial window -> pre-integrations, optimised keyframe states, trace
Run from the repository root:  python tools/make_golden_path.py"""
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
out_dir = os.path.join(ROOT, "tests", "golden")
os.makedirs(out_dir, exist_ok=True)


def kp_floats(k):
    return np.stack([k[f].astype(np.float32) for f in ("x", "y", "size", "angle", "response")] + [k["octave"].astype(np.float32)], 1)


def save(name, **kw):
    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, **kw)
    print(name, os.path.getsize(path) // 1024, "KiB")


# ---- (2) stereo ------------------------------------------------------------------------------------------------------------
left, right = synthetic.stereo_pair(21)
x0, y0, w, h = 300, 80, 480, 200
L, R = np.ascontiguousarray(left[y0:y0 + h, x0:x0 + w]), np.ascontiguousarray(right[y0:y0 + h, x0:x0 + w])
ol, orr = pyoracle.OrbOracle(nfeatures=600), pyoracle.OrbOracle(nfeatures=600)
_, kl, dl = ol.extract(L)
_, kr, dr = orr.extract(R)
bf = float(np.float32(synthetic.BF))
b = float(np.float32(synthetic.BF) / np.float32(synthetic.FX))
u, d, s = pyoracle.stereo_match(ol, orr, kl, dl, kr, dr, bf, b)
save("stereo_a", left=L, right=R, nfeatures=np.int32(600), bf=np.float32(bf), b=np.float32(b), kps_left=kp_floats(kl), desc_left=dl,
     kps_right=kp_floats(kr), desc_right=dr, u_right=u, depth=d, sad=s)

# ---- (3) LiDAR selection ---------------------------------------------------------------------------------------------------
scene = synthetic.Scene(1)
down0 = pyoracle.voxel_grid(pyoracle.lidar_preprocess(synthetic.lidar_scan(scene, 0)))
down1 = pyoracle.voxel_grid(pyoracle.lidar_preprocess(synthetic.lidar_scan(scene, 1)))
from tc2li_slam_amd import capi  # noqa: E402  (pack_lidar_state is host-side packing only, no device call)
st0, st1 = [capi.pack_lidar_state(*synthetic.lidar_state(f)[:2]) for f in (0, 1)]
boot = pyoracle.KdTree(down0[:8])
world0 = pyoracle.feature_extraction(boot, down0, st0)["world"]
body = down1[::3].copy()
tree = pyoracle.KdTree(world0)
fx = pyoracle.feature_extraction(tree, body, st1)
xyz = lambda a: np.stack([a["x"], a["y"], a["z"]], 1)  # noqa: E731
save("lidar_a", map_xyz=xyz(world0), body_xyz=xyz(body), body_intensity=body["intensity"], body_curvature=body["curvature"], state24=st1,
     selected=fx["selected"], world_xyz=xyz(fx["world"]),
     normvec=np.stack([fx["normvec"]["x"], fx["normvec"]["y"], fx["normvec"]["z"], fx["normvec"]["intensity"]], 1),
     effct_feat_num=np.int32(fx["effct_feat_num"]),
     corr_normvect=np.stack([fx["corr_normvect"][f] for f in ("x", "y", "z", "intensity")], 1))

# ---- (4) local BA with and without the LiDAR edge --------------------------------------------------------------------------
wnd = synthetic.ba_window(5, n_opt=5, n_fix=3, n_points=200, pose_noise=(0.1, 0.01))
ref = pyoracle.local_ba(wnd["poses"], wnd["fixed"], wnd["points"], wnd["edges"], wnd["cam"], iterations=10, lambda_init=0.0)
last = len(wnd["poses"]) - 1
win = np.array([last, last - 1, last - 2, last - 3], np.int32)
clouds = synthetic.ba_window_clouds(wnd, win, n_points=900)
lv = pyoracle.local_ba_lidar(wnd["poses"], wnd["fixed"], wnd["points"], wnd["edges"], wnd["cam"], win, clouds, synthetic.TCL7, 1.0)
lw = pyoracle.lidar_window_evaluate(wnd["poses"], win, clouds, synthetic.TCL7)
save("ba_a", lw_n_planes=np.int32(lw[0]), lw_residual=lw[1], lw_JacT=lw[2], lw_Hessian=lw[3], poses=wnd["poses"], fixed=wnd["fixed"], points=wnd["points"], edges=wnd["edges"], cam=wnd["cam"],
     v_poses=ref[0], v_points=ref[1], v_chi2=ref[2], v_depth_pos=ref[3], v_iterations=np.int32(ref[4]), v_trace_chi2=ref[5]["chi2"],
     v_trace_lambda=ref[5]["lam"], v_trace_trials=ref[5]["trials"],
     win=win, cloud_off=np.cumsum([0] + [len(c) for c in clouds]).astype(np.int32), clouds=np.concatenate(clouds), Tcl7=synthetic.TCL7,
     lv_poses=lv[0], lv_points=lv[1], lv_chi2=lv[2], lv_depth_pos=lv[3], lv_iterations=np.int32(lv[4]), lv_trace_chi2=lv[5]["chi2"],
     lv_trace_lambda=lv[5]["lam"], lv_trace_trials=lv[5]["trials"], lv_n_planes=np.int32(lv[6]), lv_residual=lv[7]["residual"],
     lv_JacT=lv[7]["JacT"], lv_Hessian=lv[7]["Hessian"])

# ---- (5) BALM edge on a 4-keyframe two-plane scene --------------------------------------------------------------------------
rng = np.random.default_rng(17)
W = 4
Twl = np.zeros((W, 12))
clouds2 = []
for k in range(W):
    rv = rng.normal(0, 0.02, 3)
    Rk = synthetic._rot_from_rvec(rv)
    pk = np.array([0.4 * k, 0.05 * k, 0.0]) + rng.normal(0, 0.01, 3)
    Twl[k, :9], Twl[k, 9:] = Rk.ravel(), pk
    # world points on the floor z = -1.5 and on a wall x = 6, seen from keyframe k
    n = 400
    floor = np.stack([rng.uniform(1, 5, n), rng.uniform(-2, 2, n), np.full(n, -1.5)], 1)
    wall = np.stack([np.full(n, 6.0), rng.uniform(-2, 2, n), rng.uniform(-1.4, 1.0, n)], 1)
    pw = np.concatenate([floor, wall]) + rng.normal(0, 0.01, (2 * n, 3))
    clouds2.append(((pw - pk) @ Rk).astype(np.float32))  # R^T (p - t)
n_planes, res, J, H, _ = pyoracle.balm_evaluate(Twl, clouds2)
save("balm_a", Twl=Twl, cloud_off=np.cumsum([0] + [len(c) for c in clouds2]).astype(np.int32), clouds=np.concatenate(clouds2),
     n_planes=np.int32(n_planes), residual=res, JacT=J, Hessian=H)

# ---- visual-inertial window -------------------------------------------------------------------------------------------------
iw = synthetic.inertial_window(4, n_opt=4, n_points=160)
pre298 = []
for smp, t1, t2 in iw["samples"]:
    _, f = pyoracle.imu_preintegrate(smp, t1, t2, iw["bias6"], *synthetic.IMU_NOISE)
    pre298.append(pyoracle.pack_preintegrated(f, iw["bias6"]))
pre298 = np.stack(pre298)
ir = pyoracle.local_inertial_ba(iw["kf33"], iw["fixed"], iw["has_imu"], iw["calib24"], iw["points"], iw["edges"], iw["link4"], pre298, iw["cam"])
save("inertial_a", kf33=iw["kf33"], fixed=iw["fixed"], has_imu=iw["has_imu"], calib24=iw["calib24"], points=iw["points"], edges=iw["edges"],
     link4=iw["link4"], cam=iw["cam"], bias6=iw["bias6"], noise=np.array(synthetic.IMU_NOISE),
     samples=np.concatenate([s for s, _, _ in iw["samples"]]), sample_off=np.cumsum([0] + [len(s) for s, _, _ in iw["samples"]]).astype(np.int32),
     t12=np.array([[t1, t2] for _, t1, t2 in iw["samples"]]), pre298=pre298,
     out_kf33=ir[0], out_points=ir[1], out_chi2=ir[2], out_depth_pos=ir[3], out_iterations=np.int32(ir[4]), out_trace_chi2=ir[5]["chi2"],
     out_trace_lambda=ir[5]["lam"], out_trace_trials=ir[5]["trials"], out_err=np.array(ir[6]))
