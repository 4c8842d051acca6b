"""The committed golden vectors of SURVEY.md section 8c items (2)-(5) (tests/golden/*.npz, made by tools/make_golden_path.py) and of the
local-map bookkeeping (localmap_a.npz, tools/make_golden_localmap.py) the ESKF update (eskf_a.npz, tools/make_golden_eskf.py) the tracking path (tracking_a.npz, tools/make_golden_tracking.py) the scan motion compensation (undistort_a.npz,
tools/make_golden_undistort.py) the LiDAR preprocessing + voxel filter (lidar_pre_a.npz, tools/make_golden_lidar_pre.py), the map-point refresh (mappoint_a.npz) and
LocalMapping's geometric steps (mapping_a.npz):
 - without a GPU the oracle must reproduce them (this pins the checker against silent drift);
 - on the GPU the product, called through the C ABI, is compared with the stored vectors alone -- the oracle is not involved.
Integer / byte / selection results bit for bit; optimised states within 1e-4 relative (BASELINE.json's bar)."""
import os

import numpy as np
import pytest

RTOL = 1e-4


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def kp_floats(k):
    return np.stack([k[f].astype(np.float32) for f in ("x", "y", "size", "angle", "response")] + [k["octave"].astype(np.float32)], 1)


def points(pkg_or_oracle, xyz, intensity=None, curvature=None):
    a = np.zeros(len(xyz), pkg_or_oracle.POINT_DTYPE)
    a["x"], a["y"], a["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    if intensity is not None:
        a["intensity"] = intensity
    if curvature is not None:
        a["curvature"] = curvature
    return a


def split(flat, off):
    return [np.ascontiguousarray(flat[off[i]:off[i + 1]]) for i in range(len(off) - 1)]


def rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def inertial_samples(g):
    s, off, t12 = g["samples"], g["sample_off"], g["t12"]
    return [(np.ascontiguousarray(s[off[i]:off[i + 1]]), float(t12[i, 0]), float(t12[i, 1])) for i in range(len(t12))]


# ---- the oracle against the vectors (CPU) --------------------------------------------------------------------------------
def test_oracle_stereo_golden(oracle, golden_dir):
    g = load(golden_dir, "stereo_a")
    ol, orr = oracle.OrbOracle(nfeatures=int(g["nfeatures"])), oracle.OrbOracle(nfeatures=int(g["nfeatures"]))
    _, kl, dl = ol.extract(g["left"])
    _, kr, dr = orr.extract(g["right"])
    assert np.array_equal(kp_floats(kl), g["kps_left"]) and np.array_equal(dl, g["desc_left"])
    assert np.array_equal(kp_floats(kr), g["kps_right"]) and np.array_equal(dr, g["desc_right"])
    u, d, s = oracle.stereo_match(ol, orr, kl, dl, kr, dr, float(g["bf"]), float(g["b"]))
    assert np.array_equal(u, g["u_right"]) and np.array_equal(d, g["depth"]) and np.array_equal(s, g["sad"])


def test_oracle_lidar_golden(oracle, golden_dir):
    g = load(golden_dir, "lidar_a")
    tree = oracle.KdTree(points(oracle, g["map_xyz"]))
    fx = oracle.feature_extraction(tree, points(oracle, g["body_xyz"], g["body_intensity"], g["body_curvature"]), g["state24"])
    assert np.array_equal(fx["selected"], g["selected"]) and fx["effct_feat_num"] == int(g["effct_feat_num"])
    assert np.array_equal(np.stack([fx["world"][f] for f in "xyz"], 1), g["world_xyz"])
    assert np.array_equal(np.stack([fx["normvec"][f] for f in ("x", "y", "z", "intensity")], 1), g["normvec"])
    assert np.array_equal(np.stack([fx["corr_normvect"][f] for f in ("x", "y", "z", "intensity")], 1), g["corr_normvect"])


def test_oracle_ba_golden(oracle, golden_dir):
    g = load(golden_dir, "ba_a")
    r = oracle.local_ba(g["poses"], g["fixed"], g["points"], g["edges"], g["cam"], iterations=10, lambda_init=0.0)
    assert r[4] == int(g["v_iterations"]) and np.array_equal(r[5]["trials"], g["v_trace_trials"])
    assert np.allclose(r[0], g["v_poses"], rtol=0, atol=1e-10) and np.allclose(r[1], g["v_points"], rtol=0, atol=1e-9)
    assert np.allclose(r[5]["chi2"], g["v_trace_chi2"], rtol=1e-9) and np.allclose(r[5]["lam"], g["v_trace_lambda"], rtol=1e-9)
    assert np.array_equal(r[3], g["v_depth_pos"])
    clouds = split(g["clouds"], g["cloud_off"])
    lv = oracle.local_ba_lidar(g["poses"], g["fixed"], g["points"], g["edges"], g["cam"], g["win"], clouds, g["Tcl7"], 1.0)
    assert lv[4] == int(g["lv_iterations"]) and lv[6] == int(g["lv_n_planes"]) and np.array_equal(lv[5]["trials"], g["lv_trace_trials"])
    assert np.allclose(lv[0], g["lv_poses"], rtol=0, atol=1e-10)
    assert np.allclose(lv[5]["chi2"], g["lv_trace_chi2"], rtol=1e-9) and np.allclose(lv[5]["lam"], g["lv_trace_lambda"], rtol=1e-9)
    lw = oracle.lidar_window_evaluate(g["poses"], g["win"], clouds, g["Tcl7"])
    assert lw[0] == int(g["lw_n_planes"]) and np.isclose(lw[1], float(g["lw_residual"]), rtol=1e-12)
    assert np.allclose(lw[2], g["lw_JacT"], rtol=1e-10, atol=1e-12) and np.allclose(lw[3], g["lw_Hessian"], rtol=1e-10, atol=1e-12)


def test_oracle_balm_golden(oracle, golden_dir):
    g = load(golden_dir, "balm_a")
    n, res, J, H, _ = oracle.balm_evaluate(g["Twl"], split(g["clouds"], g["cloud_off"]))
    assert n == int(g["n_planes"]) and n > 10
    assert np.isclose(res, float(g["residual"]), rtol=1e-12)
    assert np.allclose(J, g["JacT"], rtol=1e-10, atol=1e-12) and np.allclose(H, g["Hessian"], rtol=1e-10, atol=1e-12)


def test_oracle_inertial_golden(oracle, golden_dir):
    g = load(golden_dir, "inertial_a")
    pre = []
    for s, t1, t2 in inertial_samples(g):
        _, f = oracle.imu_preintegrate(s, t1, t2, g["bias6"], *g["noise"])
        pre.append(oracle.pack_preintegrated(f, g["bias6"]))
    assert np.array_equal(np.stack(pre), g["pre298"])
    r = oracle.local_inertial_ba(g["kf33"], g["fixed"], g["has_imu"], g["calib24"], g["points"], g["edges"], g["link4"], g["pre298"], g["cam"])
    assert r[4] == int(g["out_iterations"]) and np.array_equal(r[5]["trials"], g["out_trace_trials"])
    assert np.allclose(r[0], g["out_kf33"], rtol=0, atol=1e-9) and np.allclose(r[1], g["out_points"], rtol=0, atol=1e-9)
    assert np.allclose(r[6], g["out_err"], rtol=1e-9)


# ---- the product against the vectors (GPU, through the C ABI, no oracle) ---------------------------------------------------
@pytest.mark.gpu
def test_product_stereo_golden(pkg, golden_dir):
    g = load(golden_dir, "stereo_a")
    h, w = g["left"].shape
    el = pkg.OrbExtractor(nfeatures=int(g["nfeatures"]), max_width=w, max_height=h, max_images=1)
    er = pkg.OrbExtractor(nfeatures=int(g["nfeatures"]), max_width=w, max_height=h, max_images=1)
    _, kl, dl = el.extract(g["left"])
    _, kr, dr = er.extract(g["right"])
    assert np.array_equal(kp_floats(kl), g["kps_left"]) and np.array_equal(dl, g["desc_left"])
    assert np.array_equal(kp_floats(kr), g["kps_right"]) and np.array_equal(dr, g["desc_right"])
    u, d, s = pkg.compute_stereo_matches(el, er, kl, dl, kr, dr, float(g["bf"]), float(g["b"]))
    assert np.array_equal(u, g["u_right"]) and np.array_equal(d, g["depth"]) and np.array_equal(s, g["sad"])


@pytest.mark.gpu
def test_product_lidar_golden(pkg, golden_dir):
    g = load(golden_dir, "lidar_a")
    fe = pkg.LidarFrontEnd(max_points_per_scan=8192, max_scans=1)
    m = pkg.LidarMap()
    m.Build(points(pkg, g["map_xyz"]))
    fx = fe.feature_extraction(m, points(pkg, g["body_xyz"], g["body_intensity"], g["body_curvature"]), g["state24"])
    assert np.array_equal(fx["selected"], g["selected"]) and fx["effct_feat_num"] == int(g["effct_feat_num"])
    assert np.array_equal(np.stack([fx["world"][f] for f in "xyz"], 1), g["world_xyz"])
    sel = g["selected"].astype(bool)
    got = np.stack([fx["normvec"][f] for f in ("x", "y", "z", "intensity")], 1)
    assert np.array_equal(got[sel], g["normvec"][sel])
    assert np.array_equal(np.stack([fx["corr_normvect"][f] for f in ("x", "y", "z", "intensity")], 1), g["corr_normvect"])


@pytest.mark.gpu
def test_product_ba_golden(pkg, golden_dir):
    g = load(golden_dir, "ba_a")
    edges = pkg.pack_ba_edges(g["edges"])
    poses, pts, chi2, dpos, stats = pkg.local_bundle_adjustment(g["poses"], g["fixed"], g["points"], edges, g["cam"])
    assert stats.iterations == int(g["v_iterations"]) and stats.trials == int(g["v_trace_trials"].sum())
    assert rel(poses, g["v_poses"]) < RTOL and np.allclose(pts, g["v_points"], rtol=RTOL, atol=1e-5)
    assert np.array_equal(dpos, g["v_depth_pos"]) and np.allclose(chi2, g["v_chi2"], rtol=1e-3, atol=1e-4)
    clouds = split(g["clouds"], g["cloud_off"])
    n, res, J, H = pkg.capi.lidar_window_evaluate(g["poses"], g["win"], clouds, g["Tcl7"])
    assert n == int(g["lw_n_planes"]) and np.isclose(res, float(g["lw_residual"]), rtol=1e-9)
    scale = np.abs(g["lw_Hessian"]).max()
    assert np.abs(J - g["lw_JacT"]).max() <= 1e-9 * np.abs(g["lw_JacT"]).max() and np.abs(H - g["lw_Hessian"]).max() <= 1e-9 * scale
    poses, pts, chi2, dpos, stats, ls = pkg.capi.local_lv_bundle_adjustment(g["poses"], g["fixed"], g["points"], edges, g["cam"], g["win"], clouds,
                                                                           g["Tcl7"], 1.0)
    assert stats.iterations == int(g["lv_iterations"]) and stats.trials == int(g["lv_trace_trials"].sum()) and ls.n_planes == int(g["lv_n_planes"])
    assert rel(poses, g["lv_poses"]) < RTOL and np.allclose(pts, g["lv_points"], rtol=RTOL, atol=1e-5)
    assert np.array_equal(dpos, g["lv_depth_pos"])


@pytest.mark.gpu
def test_product_inertial_golden(pkg, golden_dir):
    g = load(golden_dir, "inertial_a")
    pre = []
    for i, (s, t1, t2) in enumerate(inertial_samples(g)):
        p = pkg.capi.Preintegrated(g["bias6"], *[float(x) for x in g["noise"]])
        p.preintegrate(s, t1, t2)
        f = p.fields()
        # row a11 is float arithmetic on both sides; the polar factor of NormalizeRotation differs in the last bits
        assert np.allclose(f["dR"].ravel(), g["pre298"][i, 1:10], atol=2e-6) and np.allclose(f["dP"], g["pre298"][i, 13:16], rtol=1e-4, atol=1e-6)
        pre.append(p)
    kf, pts, chi2, dpos, stats = pkg.capi.local_inertial_bundle_adjustment(g["kf33"], g["fixed"], g["has_imu"], g["calib24"], g["points"],
                                                                          pkg.pack_ba_edges(g["edges"]), g["link4"], pre, g["cam"])
    assert stats.iterations == int(g["out_iterations"]) and stats.trials == int(g["out_trace_trials"].sum())
    for k in range(len(kf)):
        assert rel(kf[k, :24], g["out_kf33"][k, :24]) < RTOL
        assert np.allclose(kf[k, 24:], g["out_kf33"][k, 24:], rtol=RTOL, atol=1e-5)
    assert np.allclose(pts, g["out_points"], rtol=RTOL, atol=1e-4)
    assert np.array_equal(dpos, g["out_depth_pos"])
    assert abs(stats.final_chi2 - g["out_err"][1]) <= 1e-3 * g["out_err"][1]


# ---- local-map bookkeeping (tools/make_golden_localmap.py) -------------------------------------------------------------------
GRAPH_KEYS = ("kf_bad", "covis_off", "covis", "child_off", "children", "parent", "prev_kf", "match_off", "matches", "point_bad", "obs_off", "obs_kf")


def _localmap_cases(g):
    graph = {k: g[k] for k in GRAPH_KEYS}
    return graph, [(g["frame_points_%d" % i], int(g["temporal_%d" % i]), g["local_kfs_%d" % i], int(g["reference_%d" % i]),
                    g["local_points_%d" % i], g["cleared_%d" % i]) for i in range(int(g["n_cases"]))]


def test_oracle_localmap_golden(oracle, golden_dir):
    graph, cases = _localmap_cases(load(golden_dir, "localmap_a"))
    assert len(cases) == 3
    for fp, temporal, kfs, ref, pts, cleared in cases:
        got = oracle.update_local_map(graph, fp, temporal)
        assert np.array_equal(got[0], kfs) and got[1] == ref and np.array_equal(got[2], pts) and np.array_equal(got[3], cleared)
        assert len(kfs) > 5 and len(pts) > 100


@pytest.mark.gpu
def test_product_localmap_golden(pkg, golden_dir):
    graph, cases = _localmap_cases(load(golden_dir, "localmap_a"))
    lm = pkg.capi.LocalMap()
    lm.set_graph(graph)
    for fp, temporal, kfs, ref, pts, cleared in cases:
        got = lm.update(fp, temporal)
        assert np.array_equal(got[0], kfs) and got[1] == ref and np.array_equal(got[2], pts) and np.array_equal(got[3], cleared)
    lm.close()


# ---- LiDAR-inertial ESKF update, row b7 (tools/make_golden_eskf.py) -----------------------------------------------------------
def _eskf_cases(g):
    return [(g["x_%d" % i], g["P_%d" % i], bool(g["ext_%d" % i]), g["out_x_%d" % i], g["out_P_%d" % i], g["out_info_%d" % i], float(g["out_res_%d" % i]))
            for i in range(int(g["n_cases"]))]


def test_oracle_eskf_golden(oracle, golden_dir):
    g = load(golden_dir, "eskf_a")
    tree = oracle.KdTree(g["map_points"])
    for xe, P, ext, want_x, want_P, info, res in _eskf_cases(g):
        xs, Ps, got = oracle.eskf_update(xe, P, tree, g["body"], max_iter=4, extrinsic_est_en=ext)
        assert [got["calls"], got["searches"], got["converged"], int(got["finished"]), got["effct_feat_num"]] == info.tolist()
        assert np.allclose(xs, want_x, rtol=1e-12, atol=1e-12) and np.allclose(np.asarray(Ps).reshape(23, 23), want_P, rtol=1e-10, atol=1e-14)
        assert abs(got["res_mean_last"] - res) <= 1e-12 * res


@pytest.mark.gpu
def test_product_eskf_golden(pkg, golden_dir):
    g = load(golden_dir, "eskf_a")
    body = g["body"]
    fe = pkg.LidarFrontEnd(max_points_per_scan=max(len(body), 256), max_scans=1)
    m = pkg.LidarMap(); m.Build(g["map_points"])
    for xe, P, ext, want_x, want_P, info, res in _eskf_cases(g):
        x, Pn, st = fe.eskf_update(m, body, xe, P, max_iter=4, extrinsic_est_en=ext)
        assert [st.calls, st.searches, st.converged, int(bool(st.finished)), st.effct_feat_num] == info.tolist()
        assert abs(st.res_mean_last - res) <= 1e-5 * res
        assert np.abs(x[:3] - want_x[:3]).max() / max(1.0, np.abs(want_x[:3]).max()) < RTOL
        assert np.abs(x[3:12] - want_x[3:12]).max() < 1e-6 and np.abs(x[24:33] - want_x[24:33]).max() < 1e-6  # rotations
        assert np.allclose(x[12:24], want_x[12:24], rtol=RTOL, atol=1e-6) and np.allclose(x[33:], want_x[33:], rtol=RTOL, atol=1e-6)
        assert np.abs(Pn - want_P).max() <= 1e-6 * np.abs(want_P).max()


# ---- pose-inertial optimisation, row a10' (tools/make_golden_pose_inertial.py) ---------------------------------------------------
def _pi_cases(g):
    for i in range(int(g["n_cases"])):
        yield (i, bool(i), g["cur33_%d" % i], g["other33_%d" % i], g["prior246_%d" % i] if i else None, g["pre298_%d" % i], g["Xw_%d" % i], g["edges_%d" % i],
               g["close_%d" % i], g["out_cur_%d" % i], g["out_other_%d" % i], g["out_outlier_%d" % i], g["out_prior_%d" % i], g["out_counts_%d" % i])


def test_oracle_pose_inertial_golden(oracle, golden_dir):
    g = load(golden_dir, "pose_inertial_a")
    for i, last, cur, oth, prior, pre, Xw, edges, close, wcur, woth, wout, wprior, wcounts in _pi_cases(g):
        got = oracle.pose_inertial(cur, oth, last, prior, g["calib24"], pre, pre, Xw, edges, close, g["cam"])
        assert np.allclose(got[0], wcur, rtol=1e-12, atol=1e-12) and np.allclose(got[1], woth, rtol=1e-12, atol=1e-12)
        assert np.array_equal(got[2], wout) and [got[4], *got[5]] == wcounts.tolist()
        assert np.allclose(got[3], wprior, rtol=1e-9, atol=1e-9 * np.abs(wprior).max())


@pytest.mark.gpu
def test_product_pose_inertial_golden(pkg, golden_dir):
    g = load(golden_dir, "pose_inertial_a")
    for i, last, cur, oth, prior, pre, Xw, edges, close, wcur, woth, wout, wprior, wcounts in _pi_cases(g):
        p = pkg.capi.Preintegrated(pre[292:298], 0.0, 0.0, 0.0, 0.0)  # the stored pre-integration, field by field
        P = p.p
        P.dT = float(pre[0])
        o = 1
        for name, k in (("dR", 9), ("dV", 3), ("dP", 3), ("JRg", 9), ("JVg", 9), ("JVa", 9), ("JPg", 9), ("JPa", 9), ("avgA", 3), ("avgW", 3), ("C", 225)):
            getattr(P, name)[:] = [float(v) for v in pre[o:o + k]]
            o += k
        got = pkg.capi.pose_inertial_optimization_batch([dict(cur33=cur, other33=oth, last_frame=last, prior246=prior, pre=p, Xw=Xw, edges=pkg.pack_ba_edges(edges),
                                                              close=close)], g["calib24"], g["cam"])[0]
        assert not got[6] and np.array_equal(got[2], wout) and [got[4], *got[5]] == wcounts.tolist()
        assert np.abs(got[0][:24] - wcur[:24]).max() / max(1.0, np.abs(wcur[:24]).max()) < RTOL and np.allclose(got[0][24:], wcur[24:], rtol=RTOL, atol=1e-6)
        assert np.abs(got[1][:24] - woth[:24]).max() / max(1.0, np.abs(woth[:24]).max()) < RTOL
        H, wH = got[3][21:].reshape(15, 15), wprior[21:].reshape(15, 15)
        assert np.abs(H - wH).max() <= 1e-4 * np.abs(wH).max()


# ---- IMU initialisation, section 8f item 4 (tools/make_golden_imu_init.py): host code on both sides, no GPU needed ------------------------
def test_oracle_imu_init_golden(oracle, golden_dir):
    g = load(golden_dir, "imu_init_a")
    n = len(g["Rwb"])
    kf = np.zeros((n, 33))
    kf[:, 12:21], kf[:, 21:24] = g["Rwb"].reshape(n, 9), g["twb"]
    vel0, Rwg0 = oracle.initial_gravity_direction(kf, g["pre298"])
    assert np.array_equal(vel0, g["vel0"]) and np.array_equal(Rwg0, g["Rwg0"])
    kf[:, 24:27] = vel0
    for tag in ("ref", "mild"):
        pg, pa = g["priors_" + tag]
        o = oracle.inertial_optimization(kf, g["pre298"], Rwg0, 1.0, np.zeros(3), np.zeros(3), priorG=float(pg), priorA=float(pa))
        assert [o[5], o[6]] == g["counts_" + tag].tolist()
        assert np.allclose(o[0][:, 24:27], g["vel_" + tag], rtol=1e-12, atol=1e-12) and np.allclose(o[1], g["Rwg_" + tag], atol=1e-13)
        assert np.allclose(o[3], g["bg_" + tag], rtol=1e-12, atol=1e-14) and np.allclose(o[4], g["ba_" + tag], rtol=1e-12, atol=1e-14)


def test_product_imu_init_golden(pkg, golden_dir):
    g = load(golden_dir, "imu_init_a")
    pres = [None]
    for pre in g["pre298"][1:]:
        p = pkg.capi.Preintegrated(pre[292:298], 0.0, 0.0, 0.0, 0.0)  # the stored pre-integration, field by field
        P = p.p
        P.dT = float(pre[0])
        o = 1
        for name, k in (("dR", 9), ("dV", 3), ("dP", 3), ("JRg", 9), ("JVg", 9), ("JVa", 9), ("JPg", 9), ("JPa", 9), ("avgA", 3), ("avgW", 3), ("C", 225)):
            getattr(P, name)[:] = [float(v) for v in pre[o:o + k]]
            o += k
        pres.append(p)
    vel0, Rwg0 = pkg.capi.imu_init_gravity(g["Rwb"], g["twb"], pres)
    assert np.allclose(vel0, g["vel0"], rtol=1e-6, atol=1e-6) and np.allclose(Rwg0, g["Rwg0"], atol=1e-6)
    gdir = lambda R: np.asarray(R, np.float64) @ [0, 0, -1.0]
    assert np.degrees(np.arccos(np.clip(gdir(Rwg0) @ gdir(g["Rwg_true"]), -1, 1))) < 3.0
    for tag in ("ref", "mild"):
        pg, pa = g["priors_" + tag]
        v, R, s, bg, ba, st = pkg.capi.inertial_optimization(g["Rwb"], g["twb"], g["vel0"], pres, g["Rwg0"], 1.0, np.zeros(3), np.zeros(3), prior_g=float(pg),
                                                           prior_a=float(pa))
        assert abs(st.iterations - int(g["counts_" + tag][0])) <= 2 and abs(st.final_chi2 - g["err_" + tag][1]) <= 1e-5 * g["err_" + tag][1]
        assert np.allclose(v, g["vel_" + tag], rtol=1e-4, atol=1e-4) and np.allclose(R, g["Rwg_" + tag], atol=1e-5) and s == 1.0
        # with the reference's priors the accelerometer bias keeps creeping while trials at lambda ~ 1e8 are still accepted (the prior edge's
        # gradient has the wrong sign, see tests/test_imu_init.py): how many are is decided by rounding, hence the absolute tolerance
        assert np.allclose(bg, g["bg_" + tag], rtol=1e-4, atol=1e-6) and np.allclose(ba, g["ba_" + tag], rtol=1e-4, atol=5e-4 if tag == "ref" else 1e-4)
        assert np.abs(bg - g["bg_true"]).max() < 1e-3 and np.degrees(np.arccos(np.clip(gdir(R) @ gdir(g["Rwg_true"]), -1, 1))) < 5.0


# ---- tracking data path, rows a9 + a10 composed (tools/make_golden_tracking.py) -------------------------------------------------
def _keys_from_floats(dtype, a):
    k = np.zeros(len(a), dtype)
    for i, f in enumerate(("x", "y", "size", "angle", "response")):
        k[f] = a[:, i]
    k["octave"] = a[:, 5].astype(np.int32)
    return k


def test_oracle_tracking_golden(oracle, golden_dir):
    g = load(golden_dir, "tracking_a")
    h, w = g["left"].shape
    n = int(g["nfeatures"])
    ol, orr = oracle.OrbOracle(nfeatures=n), oracle.OrbOracle(nfeatures=n)
    _, kl, dl = ol.extract(g["left"])
    _, kr, dr = orr.extract(g["right"])
    u, d, _ = oracle.stereo_match(ol, orr, kl, dl, kr, dr, float(g["bf"]), float(g["b"]))
    lk = _keys_from_floats(kl.dtype, g["last_keys"])
    r = oracle.track_motion_model(kl, dl, u, w, h, g["scales"], g["inv_sigma2"], g["pred7"], g["last_pose7"], g["cam5"], float(g["b"]), float(g["th"]),
                                  g["last_has_point"], g["last_outlier"], g["last_Xw"], lk, g["last_desc"])
    assert int(r[2]) == int(g["out_n_matches"]) and int(r[3]) == int(g["out_n_inliers"]) and int(r[3]) > 200
    assert np.array_equal(np.asarray(r[1], np.int32), g["out_matches"]) and np.allclose(r[0], g["out_pose7"], rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
def test_product_tracking_golden(pkg, golden_dir):
    import torch
    g = load(golden_dir, "tracking_a")
    h, w = g["left"].shape
    ext = pkg.OrbExtractor(nfeatures=int(g["nfeatures"]), max_width=w, max_height=h, max_images=2)
    dev = torch.from_numpy(np.stack([g["left"], g["right"]])).cuda()
    kps, desc, counts, _ = ext.extract_batch_dev(dev.data_ptr(), 2, w, h, w, w * h)
    u_right, depth, _ = pkg.stereo_match_batch(ext, 1, float(g["bf"]), float(g["b"]))
    last = dict(has_point=g["last_has_point"], outlier=g["last_outlier"], Xw=g["last_Xw"], keys=_keys_from_floats(kps.dtype, g["last_keys"]),
                descriptors=g["last_desc"], pose7=g["last_pose7"])
    poses, mp, nm, inl = pkg.capi.track_motion_model_batch(ext, 1, kps, u_right, pkg.capi.pack_last_frames([last]), g["pred7"][None], g["cam5"],
                                                            float(g["b"]), float(g["th"]))
    n = int(counts[0])
    assert n == len(g["out_matches"]) and int(nm[0]) == int(g["out_n_matches"]) and int(inl[0]) == int(g["out_n_inliers"])
    assert np.array_equal(mp[0, :n], g["out_matches"])
    assert np.allclose(poses[0], g["out_pose7"], rtol=RTOL, atol=1e-6)
    ext.close()


# ---- scan motion compensation, row b2 (tools/make_golden_undistort.py) ----------------------------------------------------------
def _lidar_state24(st36):
    return np.concatenate([st36[3:12], st36[0:3], st36[24:33], st36[33:36]])


def test_oracle_undistort_golden(oracle, golden_dir):
    g = load(golden_dir, "undistort_a")
    t = g["t"]
    st, poses = oracle.imu_propagate(g["state0"], g["imu"], t[0], t[1], t[2], t[3], g["last6"])
    assert np.allclose(st, g["out_state"], rtol=1e-13, atol=1e-13) and np.allclose(poses, g["out_poses"], rtol=1e-13, atol=1e-13)
    out = oracle.undistort(g["points"], poses, _lidar_state24(st))
    assert out.tobytes() == g["out_points"].tobytes() and len(out) > 5000


@pytest.mark.gpu
def test_product_undistort_golden(pkg, golden_dir):
    g = load(golden_dir, "undistort_a")
    t = g["t"]
    st, poses, _ = pkg.capi.lidar_imu_propagate(g["state0"], g["imu"], t[0], t[1], t[2], t[3], g["last6"])
    assert np.allclose(st, g["out_state"], rtol=1e-12, atol=1e-12) and np.allclose(poses, g["out_poses"], rtol=1e-12, atol=1e-12)
    fe = pkg.LidarFrontEnd(max_points_per_scan=len(g["points"]) + 256, max_scans=1)
    got, want = fe.undistort(g["points"], g["out_poses"], _lidar_state24(g["out_state"])), g["out_points"]
    for name in ("intensity", "curvature", "normal_x"):  # identical order: the other fields travel with the point
        assert np.array_equal(got[name], want[name]), name
    xyz_g, xyz_w = np.stack([got["x"], got["y"], got["z"]], 1), np.stack([want["x"], want["y"], want["z"]], 1)
    assert np.all(np.abs(xyz_g - xyz_w) <= np.spacing(np.abs(xyz_w).astype(np.float32)))  # double results rounded to float: one ulp at most


# ---- LiDAR preprocessing + voxel filter, rows b1 + b3 (tools/make_golden_lidar_pre.py) -------------------------------------------
def test_oracle_lidar_pre_golden(oracle, golden_dir):
    g = load(golden_dir, "lidar_pre_a")
    pre = oracle.lidar_preprocess(g["raw"], int(g["point_filter_num"]), float(g["blind"]), float(g["time_unit_scale"]))
    assert pre.tobytes() == g["out_pre"].tobytes() and len(pre) > 5000
    down = oracle.voxel_grid(pre, float(g["leaf"]))
    assert down.tobytes() == g["out_down"].tobytes() and 1000 < len(down) < len(pre)


@pytest.mark.gpu
def test_product_lidar_pre_golden(pkg, golden_dir):
    g = load(golden_dir, "lidar_pre_a")
    fe = pkg.LidarFrontEnd(max_points_per_scan=len(g["raw"]) + 256, max_scans=1)
    pre = fe.process(g["raw"], int(g["point_filter_num"]), float(g["blind"]), float(g["time_unit_scale"]))
    assert pre.tobytes() == g["out_pre"].tobytes()
    down = fe.voxel_filter(g["out_pre"], float(g["leaf"]))
    assert down.tobytes() == g["out_down"].tobytes()


# ---- map-point refresh, section 8f item 3 (tools/make_golden_mappoint.py) ----------------------------------------------------------
def _mappoint_args(g):
    return g["obs_off"], g["descriptors"], g["centres"], g["positions"], g["ref_centres"], g["level_scale"], float(g["last_scale"])


def _check_mappoint(got, g):
    assert np.array_equal(got[0], g["out_best"])
    has = g["out_best"] >= 0
    assert has.sum() > 150
    for a, name in zip(got[1:], ("out_normals", "out_min", "out_max")):
        assert np.array_equal(a[has], g[name][has]), name  # float arithmetic in the reference's order: bit for bit


def test_oracle_mappoint_golden(oracle, golden_dir):
    g = load(golden_dir, "mappoint_a")
    _check_mappoint(oracle.map_points_refresh(*_mappoint_args(g)), g)


@pytest.mark.gpu
def test_product_mappoint_golden(pkg, golden_dir):
    g = load(golden_dir, "mappoint_a")
    _check_mappoint(pkg.capi.map_points_refresh(*_mappoint_args(g)), g)


# ---- LocalMapping's geometric steps, section 8f item 1 (tools/make_golden_mapping.py) ----------------------------------------------
def _mapping_keyframes(g, key_dtype):
    kfs = []
    for i in range(int(g["n_kf"])):
        kfs.append(dict(keys=_keys_from_floats(key_dtype, g["keys_%d" % i]), descriptors=g["desc_%d" % i], u_right=g["u_right_%d" % i],
                        depth=g["depth_%d" % i], has_point=g["has_point_%d" % i], fv_node=g["fv_node_%d" % i], fv_offset=g["fv_offset_%d" % i],
                        fv_index=g["fv_index_%d" % i], pose7=g["pose7_%d" % i], centre=g["centre_%d" % i]))
    return kfs


def _check_mapping(tri, new, fuse, g):
    assert tri[0] == int(g["out_tri_n"]) and np.array_equal(tri[1], g["out_tri_matches"]) and tri[0] > 20
    assert np.array_equal(new[0], g["out_new_idx"]) and len(new[0]) > 20
    stereo = g["out_new_idx"][:, 3] == 1
    assert np.array_equal(new[1][stereo], g["out_new_x3"][stereo])                        # un-projections: bit for bit
    assert np.allclose(new[1][~stereo], g["out_new_x3"][~stereo], rtol=RTOL, atol=1e-5)   # triangulations: float rounding of the eigenvector
    assert fuse[0] == int(g["out_fuse_n"]) and np.array_equal(fuse[1], g["out_fuse_idx"]) and np.array_equal(fuse[2], g["out_fuse_dist"])


def test_oracle_mapping_golden(oracle, golden_dir):
    g = load(golden_dir, "mapping_a")
    kfs = _mapping_keyframes(g, oracle.OrbOracle(nfeatures=10).extract(np.zeros((64, 64), np.uint8))[1].dtype)
    cam4, mb, mbf, sf, sg = g["cam4"], float(g["mb"]), float(g["mbf"]), g["sf"], g["sg"]
    B = kfs[0]
    _check_mapping(oracle.search_for_triangulation(kfs[0], kfs[2], cam4, sf, sg), oracle.create_new_map_points(kfs[0], kfs[1:], cam4, mb, mbf, sf, sg),
                   oracle.fuse_search(B["keys"], B["descriptors"], B["u_right"], int(g["width"]), int(g["height"]), B["pose7"], cam4, mbf, sf, g["fuse_isg"],
                                      float(g["fuse_logsf"]), g["fuse_points"], g["fuse_valid"], th=3.0), g)


@pytest.mark.gpu
def test_product_mapping_golden(pkg, golden_dir):
    g = load(golden_dir, "mapping_a")
    kfs = _mapping_keyframes(g, pkg.capi.KEYPOINT_DTYPE)
    cam4, mb, mbf, sf, sg = g["cam4"], float(g["mb"]), float(g["mbf"]), g["sf"], g["sg"]
    cam5 = np.float32([cam4[0], cam4[1], cam4[2], cam4[3], mbf]).astype(np.float64)
    B = kfs[0]
    _check_mapping(pkg.capi.search_for_triangulation(kfs[0], kfs[2], cam5, sf, sg), pkg.capi.create_new_map_points(kfs[0], kfs[1:], cam5, mb, sf, sg),
                   pkg.capi.fuse_search(B["keys"], B["descriptors"], B["u_right"], int(g["width"]), int(g["height"]), B["pose7"], cam4, mbf, sf, g["fuse_isg"],
                                        float(g["fuse_logsf"]), g["fuse_points"], g["fuse_valid"], th=3.0), g)
