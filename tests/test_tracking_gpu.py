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
