"""Shared scenario for the projection-matching tests: a current frame (ORB + stereo from the oracle) and source
points derived from it (back-projected stereo keypoints, perturbed descriptors, a slightly different camera pose)."""
import numpy as np


def make(oracle, synthetic, seed=0, w=1242, h=375, nfeat=2000, flips=24, motion=0.05):
    rng = np.random.default_rng(seed)
    left, right = synthetic.stereo_pair(seed, w, h)
    ol, orr = oracle.OrbOracle(nfeatures=nfeat), oracle.OrbOracle(nfeatures=nfeat)
    _, kl, dl = ol.extract(left)
    _, kr, dr = orr.extract(right)
    bf = np.float32(synthetic.BF)
    b = np.float32(bf / np.float32(synthetic.FX))
    u_right, depth, _ = oracle.stereo_match(ol, orr, kl, dl, kr, dr, float(bf), float(b))
    scales = ol.tables()[0]
    fx, fy, cx, cy = [np.float32(v) for v in (synthetic.FX, synthetic.FY, synthetic.CX, synthetic.CY)]
    n = len(kl)
    # "last frame": the same keypoints in a shuffled order with 3-D points from stereo (camera frame of the last pose = world)
    order = rng.permutation(n)
    last_keys = kl[order].copy()
    last_keys["angle"] = (last_keys["angle"] + rng.normal(0, 3, n).astype(np.float32)) % np.float32(360)
    z = depth[order]
    has_point = (z > 0).astype(np.uint8)
    zz = np.where(z > 0, z, 1).astype(np.float32)
    Xw = np.stack([(last_keys["x"] - cx) * zz / fx, (last_keys["y"] - cy) * zz / fy, zz], 1).astype(np.float32)
    outlier = (rng.random(n) < 0.05).astype(np.uint8)
    mp_desc = dl[order].copy()
    for i in range(n):  # descriptor drift between frames
        bits = rng.choice(256, size=int(rng.integers(0, flips)), replace=False)
        for bit in bits:
            mp_desc[i, bit // 8] ^= np.uint8(1 << (bit % 8))
    pose_last = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
    ang = motion * 0.02
    pose_cur = np.array([0, np.sin(ang / 2), 0, np.cos(ang / 2), 0.02, -0.01, -motion], np.float32)  # moved forward
    occupied = (rng.random(n) < 0.03).astype(np.uint8)
    return dict(keys=kl, desc=dl, u_right=u_right, depth=depth, scales=scales, cols=w, rows=h, cam4=np.array([fx, fy, cx, cy], np.float32),
                bf=float(bf), b=float(b), last_keys=last_keys, has_point=has_point, outlier=outlier, Xw=Xw, mp_desc=mp_desc,
                pose_last=pose_last, pose_cur=pose_cur, occupied=occupied, order=order)


def local_map_points(sc, oracle, rng, n_extra=500):
    """Local map points: the stereo points of the frame plus far / behind / oblique extras."""
    n = len(sc["last_keys"])
    pts = np.zeros(n + n_extra, oracle.MAP_POINT_DTYPE)
    pts["pos"][:n] = sc["Xw"]
    pts["pos"][n:] = rng.uniform([-30, -5, -5], [30, 3, 80], (n_extra, 3))
    d = np.linalg.norm(pts["pos"], axis=1).astype(np.float32)
    view = pts["pos"] / np.maximum(d[:, None], 1e-3)
    pts["normal"] = (view + rng.normal(0, 0.2, view.shape)).astype(np.float32)
    pts["normal"] /= np.linalg.norm(pts["normal"], axis=1, keepdims=True)
    oct_ = np.concatenate([sc["last_keys"]["octave"], rng.integers(0, 8, n_extra)])
    pts["max_distance_raw"] = d * sc["scales"][oct_]
    pts["max_distance"] = np.float32(1.2) * pts["max_distance_raw"]
    pts["min_distance"] = np.float32(0.8) * pts["max_distance_raw"] / sc["scales"][7]
    pts["descriptor"][:n] = sc["mp_desc"]
    pts["descriptor"][n:] = rng.integers(0, 256, (n_extra, 32))
    keep = np.concatenate([sc["has_point"] > 0, np.ones(n_extra, bool)])
    return pts[keep]
