"""GPU parity of the batched TrackWithMotionModel data path (projection matching on the device-resident features ->
pose-only optimisation) with the oracle's composition of the same reference functions, frame by frame."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def scenario(pkg, synthetic, seeds, w=1242, h=375, variants=None):
    import torch
    F = len(seeds)
    imgs = np.stack([np.stack(synthetic.stereo_pair(s, w, h)) for s in seeds]).reshape(2 * F, h, w)
    dev = torch.from_numpy(imgs).cuda()
    ext = pkg.OrbExtractor(max_width=w, max_height=h, max_images=2 * F)
    kps, desc, counts, _ = ext.extract_batch_dev(dev.data_ptr(), 2 * F, w, h, w, w * h)
    bf = np.float32(synthetic.BF); b = np.float32(bf / np.float32(synthetic.FX))
    u_right, depth, _ = pkg.stereo_match_batch(ext, F, float(bf), float(b))
    fx, fy, cx, cy = [np.float32(v) for v in (synthetic.FX, synthetic.FY, synthetic.CX, synthetic.CY)]
    lasts, preds = [], []
    for f in range(F):
        rng = np.random.default_rng(100 + f)
        n = int(counts[2 * f])
        kl, dl = kps[2 * f, :n], desc[2 * f, :n]
        order = rng.permutation(n)
        variant = (variants or {}).get(f, "normal")
        if variant == "few":      # so few points that even the wide window finds < 20
            order = order[:12]
        elif variant == "retry":  # a large prediction error: the narrow window fails, the wide one succeeds
            pass
        m = len(order)
        lk = kl[order].copy()
        lk["angle"] = (lk["angle"] + rng.normal(0, 3, m).astype(np.float32)) % np.float32(360)
        z = depth[f, :n][order]
        has_point = (z > 0).astype(np.uint8)
        zz = np.where(z > 0, z, 1).astype(np.float32)
        Xw = np.stack([(lk["x"] - cx) * zz / fx, (lk["y"] - cy) * zz / fy, zz], 1).astype(np.float32)
        outlier = (rng.random(m) < 0.05).astype(np.uint8)
        md = dl[order].copy()
        for i in range(m):
            for bit in rng.choice(256, size=int(rng.integers(0, 20)), replace=False):
                md[i, bit // 8] ^= np.uint8(1 << (bit % 8))
        lasts.append(dict(has_point=has_point, outlier=outlier, Xw=Xw, keys=lk, descriptors=md, pose7=np.array([0, 0, 0, 1, 0, 0, 0], np.float32)))
        ang = 0.001 if variant != "retry" else 0.012
        preds.append(np.array([0, np.sin(ang / 2), 0, np.cos(ang / 2), 0.02, -0.01, -0.05], np.float32))
    return dict(ext=ext, dev=dev, kps=kps, desc=desc, counts=counts, u_right=u_right, lasts=lasts, preds=np.stack(preds), bf=float(bf), b=float(b),
                w=w, h=h)


def check(pkg, oracle, synthetic, sc, th=7.0):
    F = len(sc["lasts"])
    cam5 = np.array([synthetic.FX, synthetic.FY, synthetic.CX, synthetic.CY, sc["bf"]], np.float64)
    cam5 = np.float32(cam5).astype(np.float64)  # the frame's intrinsics are floats
    packed = pkg.capi.pack_last_frames(sc["lasts"])
    poses, mp, nm, inl = pkg.capi.track_motion_model_batch(sc["ext"], F, sc["kps"], sc["u_right"], packed, sc["preds"], cam5, sc["b"], th)
    scales, inv_sigma2 = sc["ext"].GetScaleFactors(), sc["ext"].GetInverseScaleSigmaSquares()
    out = []
    for f in range(F):
        n = int(sc["counts"][2 * f])
        L = sc["lasts"][f]
        want = oracle.track_motion_model(sc["kps"][2 * f, :n], sc["desc"][2 * f, :n], sc["u_right"][f, :n], sc["w"], sc["h"], scales, inv_sigma2,
                                         sc["preds"][f], L["pose7"], cam5, sc["b"], th, L["has_point"], L["outlier"], L["Xw"], L["keys"],
                                         L["descriptors"])
        assert nm[f] == want[2]
        assert inl[f] == want[3]
        assert np.array_equal(mp[f, :n], want[1])
        assert np.all(mp[f, n:] == -1)
        assert np.allclose(poses[f], want[0], rtol=1e-4, atol=1e-6)
        out.append((nm[f], inl[f]))
    return out


def test_track_motion_model_batch(pkg, oracle, synthetic):
    sc = scenario(pkg, synthetic, [0, 1, 2, 3, 4, 5])
    res = check(pkg, oracle, synthetic, sc)
    assert all(nm > 300 and inl > 200 for nm, inl in res)


def test_track_motion_model_retry_and_failure(pkg, oracle, synthetic):
    sc = scenario(pkg, synthetic, [6, 7, 8, 9], variants={1: "few", 2: "retry"})
    res = check(pkg, oracle, synthetic, sc, th=2.0)
    assert res[1][1] == -1 and res[1][0] < 20      # tracking lost: PoseOptimization not run
    assert res[0][1] > 50 and res[3][1] > 50


def test_track_motion_model_argument_errors(pkg, synthetic):
    sc = scenario(pkg, synthetic, [0])
    cam5 = np.array([synthetic.FX, synthetic.FY, synthetic.CX, synthetic.CY, sc["bf"]])
    packed = pkg.capi.pack_last_frames(sc["lasts"])
    with pytest.raises(pkg.capi.Tc2liError):  # more frames than the extractor holds
        pkg.capi.track_motion_model_batch(sc["ext"], 2, np.tile(sc["kps"], (2, 1)), np.tile(sc["u_right"], (2, 1)),
                                          pkg.capi.pack_last_frames(sc["lasts"] * 2), np.tile(sc["preds"], (2, 1)), cam5, sc["b"])
    del packed


# ---- Tracking::TrackLocalMap --------------------------------------------------------------------------------------------------
def local_map_scenario(pkg, oracle, synthetic, seeds):
    sc = scenario(pkg, synthetic, seeds)
    F = len(seeds)
    cap = sc["kps"].shape[1]
    fx, fy, cx, cy = [np.float32(v) for v in (synthetic.FX, synthetic.FY, synthetic.CX, synthetic.CY)]
    sf = sc["ext"].GetScaleFactors()
    held, held_Xw, pts_all, offs, poses = np.zeros((F, cap), np.uint8), np.zeros((F, cap, 3), np.float32), [], [0], []
    _, depth, _ = pkg.stereo_match_batch(sc["ext"], F, sc["bf"], sc["b"])
    for f in range(F):
        rng = np.random.default_rng(500 + f)
        n = int(sc["counts"][2 * f])
        k, d, z = sc["kps"][2 * f, :n], sc["desc"][2 * f, :n], depth[f, :n]
        ok = z > 0
        zz = np.where(ok, z, 1).astype(np.float32)
        X = np.stack([(k["x"] - cx) * zz / fx, (k["y"] - cy) * zz / fy, zz], 1).astype(np.float32)
        roll = rng.random(n)
        h = np.where(ok & (roll < 0.3), 1, np.where(ok & (roll < 0.35), 2, 0)).astype(np.uint8)
        held[f, :n], held_Xw[f, :n] = h, X
        loc = np.nonzero(ok & (h == 0) & (roll < 0.9))[0]
        loc = loc[rng.permutation(len(loc))]
        pts = np.zeros(len(loc), pkg.MAP_POINT_DTYPE)
        pts["pos"] = X[loc] + rng.normal(0, 0.01, (len(loc), 3)).astype(np.float32)
        dist = np.linalg.norm(X[loc], axis=1).astype(np.float32)
        pts["normal"] = X[loc] / dist[:, None]
        raw = (dist * sf[k["octave"][loc]]).astype(np.float32)
        pts["max_distance_raw"], pts["max_distance"], pts["min_distance"] = raw, np.float32(1.2) * raw, np.float32(0.8) * (raw / sf[-1])
        md = d[loc].copy()
        for i in range(len(loc)):
            for bit in rng.choice(256, size=int(rng.integers(0, 25)), replace=False):
                md[i, bit // 8] ^= np.uint8(1 << (bit % 8))
        pts["descriptor"] = md
        pts_all.append(pts)
        offs.append(offs[-1] + len(pts))
        ang = 0.0015
        poses.append(np.array([0, np.sin(ang / 2), 0, np.cos(ang / 2), 0.01, -0.005, -0.03], np.float32))
    sc.update(held=held, held_Xw=held_Xw, local=np.concatenate(pts_all), local_off=np.array(offs, np.int32), poses=np.stack(poses))
    return sc


@pytest.mark.parametrize("th,far", [(1.0, False), (6.0, False), (2.0, True)])
def test_track_local_map_batch(pkg, oracle, synthetic, th, far):
    sc = local_map_scenario(pkg, oracle, synthetic, [10, 11, 12, 13])
    F = len(sc["poses"])
    cam5 = np.float32([synthetic.FX, synthetic.FY, synthetic.CX, synthetic.CY, sc["bf"]]).astype(np.float64)
    got = pkg.capi.track_local_map_batch(sc["ext"], F, sc["kps"], sc["u_right"], sc["poses"], sc["held"], sc["held_Xw"], sc["local"], sc["local_off"], cam5,
                                         th=th, far_points=far, th_far=30.0)
    scales, inv_sigma2 = sc["ext"].GetScaleFactors(), sc["ext"].GetInverseScaleSigmaSquares()
    log_scale = float(np.log(np.float32(1.2)))
    for f in range(F):
        n = int(sc["counts"][2 * f])
        pts = sc["local"][sc["local_off"][f]:sc["local_off"][f + 1]]
        want = oracle.track_local_map(sc["kps"][2 * f, :n], sc["desc"][2 * f, :n], sc["u_right"][f, :n], sc["w"], sc["h"], scales, inv_sigma2, log_scale,
                                      sc["poses"][f], cam5, sc["held"][f, :n], sc["held_Xw"][f, :n], pts, th=th, far_points=far, th_far=30.0)
        assert got[3][f] == want[3] and want[3] > 100
        assert np.array_equal(got[1][f, :n], want[1]) and np.all(got[1][f, n:] == -1)
        assert np.array_equal(got[2][f, :n], want[2])
        assert got[4][f] == want[4] and want[4] > 200
        assert np.allclose(got[0][f], want[0], rtol=1e-4, atol=1e-6)
        # a keypoint that holds a point with observations keeps it
        assert np.all(got[1][f, :n][sc["held"][f, :n] == 1] == -1)


@pytest.mark.parametrize("th,far", [(1.0, False), (2.0, True)])
def test_search_local_points_batch(pkg, oracle, synthetic, th, far):
    """tc2li_search_local_points_batch = Tracking::SearchLocalPoints alone (the inertial configuration's TrackLocalMap optimises with
    PoseInertialOptimization, Tracking.cc:2857-2878): the keypoints' new map points and the match counts of the oracle's TrackLocalMap."""
    sc = local_map_scenario(pkg, oracle, synthetic, [10, 11, 12, 13])
    F = len(sc["poses"])
    cam5 = np.float32([synthetic.FX, synthetic.FY, synthetic.CX, synthetic.CY, sc["bf"]]).astype(np.float64)
    lk, nm = pkg.capi.search_local_points_batch(sc["ext"], F, sc["kps"], sc["u_right"], sc["poses"], sc["held"], sc["held_Xw"], sc["local"], sc["local_off"], cam5,
                                                th=th, far_points=far, th_far=30.0)
    scales, inv_sigma2 = sc["ext"].GetScaleFactors(), sc["ext"].GetInverseScaleSigmaSquares()
    log_scale = float(np.log(np.float32(1.2)))
    for f in range(F):
        n = int(sc["counts"][2 * f])
        pts = sc["local"][sc["local_off"][f]:sc["local_off"][f + 1]]
        want = oracle.track_local_map(sc["kps"][2 * f, :n], sc["desc"][2 * f, :n], sc["u_right"][f, :n], sc["w"], sc["h"], scales, inv_sigma2, log_scale,
                                      sc["poses"][f], cam5, sc["held"][f, :n], sc["held_Xw"][f, :n], pts, th=th, far_points=far, th_far=30.0)
        assert nm[f] == want[3] and want[3] > 100
        assert np.array_equal(lk[f, :n], want[1]) and np.all(lk[f, n:] == -1)
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.search_local_points_batch(sc["ext"], F, sc["kps"], sc["u_right"], sc["poses"], sc["held"], sc["held_Xw"], sc["local"],
                                           np.array([1, 5, 9, 12, 20], np.int32), cam5)


def test_track_local_map_edge_cases(pkg, oracle, synthetic):
    sc = local_map_scenario(pkg, oracle, synthetic, [14, 15])
    cam5 = np.float32([synthetic.FX, synthetic.FY, synthetic.CX, synthetic.CY, sc["bf"]]).astype(np.float64)
    # no local points at all: the pose is optimised over the held points only
    got = pkg.capi.track_local_map_batch(sc["ext"], 2, sc["kps"], sc["u_right"], sc["poses"], sc["held"], sc["held_Xw"], sc["local"][:0],
                                         np.zeros(3, np.int32), cam5)
    assert np.all(got[3] == 0) and np.all(got[1] == -1) and np.all(got[4] > 50)
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.track_local_map_batch(sc["ext"], 2, sc["kps"], sc["u_right"], sc["poses"], sc["held"], sc["held_Xw"], sc["local"],
                                       np.array([1, 5, 9], np.int32), cam5)


def test_matcher_pool_overflow_takes_the_one_kernel_path(pkg, oracle, synthetic, monkeypatch):
    """With a candidate pool of one entry per query the list form overflows and the batch falls back to the one-kernel matcher:
    the results must not change."""
    sc = scenario(pkg, synthetic, [20, 21])
    ref = check(pkg, oracle, synthetic, sc)
    monkeypatch.setenv("TC2LI_MATCH_POOL_PER_QUERY", "1")
    assert check(pkg, oracle, synthetic, sc) == ref
