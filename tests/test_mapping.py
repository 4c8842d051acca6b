"""CreateNewMapPoints core: ORBmatcher::SearchForTriangulation and the pair loop of
LocalMapping::CreateNewMapPoints.  CPU: properties of the oracle on keyframes rendered along a synthetic drive (geometric
consistency of the matches, the triangulated points against the known scene).  GPU: the product through the C ABI against the
oracle -- identical matches and identical sets of created points, coordinates within float rounding of the triangulation."""
import numpy as np
import pytest

W, H = 800, 300
NFEAT = 1200


def camera_position(k):
    return np.array([0.25 * k + 0.05 * np.sin(1.3 * k), 0.0, 0.45 * k])


def make_keyframes(synthetic, extract, stereo, n_kf, seed=5, mono_fraction=0.3, with_points=0.45):
    """extract(img) -> (keys, desc) per image pair side; stereo(k, ...) -> (u_right, depth).  Returns keyframe dicts, newest first."""
    scene = synthetic.Scene(seed)
    rng = np.random.default_rng(100 + seed)
    kfs = []
    for k in range(n_kf):
        c = camera_position(k)
        left, _ = scene.render(c[0], W, H, noise_seed=2 * k + 1, cam_z=c[2])
        right, _ = scene.render(c[0] + synthetic.BASELINE, W, H, noise_seed=2 * k + 2, cam_z=c[2])
        keys, desc, u_right, depth = extract(left, right)
        n = len(keys)
        drop = rng.random(n) < mono_fraction  # some keypoints without a stereo match (mono observations)
        u_right = np.where(drop, np.float32(-1), u_right).astype(np.float32)
        depth = np.where(drop, np.float32(-1), depth).astype(np.float32)
        node = (desc[:, 0] & 1).astype(np.int32) | ((desc[:, 5] & 1) << 1) | ((desc[:, 9] & 1) << 2) | ((desc[:, 14] & 1) << 3) | \
               ((desc[:, 21] & 1).astype(np.int32) << 4) | ((desc[:, 27] & 1).astype(np.int32) << 5)
        node = node * 7 + 3  # non-contiguous ids
        ids = np.unique(node)
        order = np.argsort(node, kind="stable")
        off = np.concatenate([[0], np.cumsum([(node == i).sum() for i in ids])]).astype(np.int32)
        kfs.append(dict(keys=keys, descriptors=desc, u_right=u_right, depth=depth, has_point=(rng.random(n) < with_points).astype(np.uint8),
                        fv_node=ids.astype(np.int32), fv_offset=off, fv_index=order.astype(np.int32),
                        pose7=np.concatenate([[0, 0, 0, 1], -c]).astype(np.float32), centre=c))
    return kfs[::-1]


def oracle_features(oracle, synthetic):
    ol, orr = oracle.OrbOracle(nfeatures=NFEAT), oracle.OrbOracle(nfeatures=NFEAT)
    bf = float(np.float32(synthetic.BF)); b = float(np.float32(synthetic.BF) / np.float32(synthetic.FX))

    def extract(left, right):
        _, kl, dl = ol.extract(left)
        _, kr, dr = orr.extract(right)
        u, d, _ = oracle.stereo_match(ol, orr, kl, dl, kr, dr, bf, b)
        return kl, dl, u, d
    return extract


def tables(n_levels=8):
    sf = (np.float32(1.2) ** np.arange(n_levels)).astype(np.float32)
    sf = np.cumprod(np.concatenate([[np.float32(1)], np.full(n_levels - 1, np.float32(1.2))])).astype(np.float32)  # mvScaleFactors[i] = [i-1] * 1.2
    return sf, (sf * sf).astype(np.float32)


@pytest.fixture(scope="module")
def keyframes(oracle, synthetic):
    return make_keyframes(synthetic, oracle_features(oracle, synthetic), None, 5)


def cam_of(synthetic):
    bf = np.float32(synthetic.BF)
    return np.float32([synthetic.FX, synthetic.FY, synthetic.CX, synthetic.CY]), float(bf), float(bf / np.float32(synthetic.FX))


def test_oracle_search_for_triangulation(oracle, synthetic, keyframes):
    cam4, mbf, mb = cam_of(synthetic)
    sf, sg = tables()
    cur, nb = keyframes[0], keyframes[2]
    n, m = oracle.search_for_triangulation(cur, nb, cam4, sf, sg)
    assert n == (m >= 0).sum() and n > 40
    sel = np.nonzero(m >= 0)[0]
    assert not cur["has_point"][sel].any() and not nb["has_point"][m[sel]].any()
    # matched keypoints lie in the same vocabulary node
    node1 = np.zeros(len(cur["keys"]), np.int32); node2 = np.zeros(len(nb["keys"]), np.int32)
    for kf, node in ((cur, node1), (nb, node2)):
        for a in range(len(kf["fv_node"])):
            node[kf["fv_index"][kf["fv_offset"][a]:kf["fv_offset"][a + 1]]] = kf["fv_node"][a]
    assert np.array_equal(node1[sel], node2[m[sel]])
    # pure translation between the views: the epipolar lines pass through the epipole; matched points lie near theirs
    t12 = (cur["centre"] - nb["centre"]) * -1.0  # t of T12 = T1w * Tw2 = c2 - c1 ... sign irrelevant for the line distance
    K = np.array([[cam4[0], 0, cam4[2]], [0, cam4[1], cam4[3]], [0, 0, 1]], np.float64)
    tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    F = np.linalg.inv(K).T @ tx @ np.linalg.inv(K)
    x1 = np.stack([cur["keys"]["x"][sel], cur["keys"]["y"][sel], np.ones(len(sel))], 1)
    x2 = np.stack([nb["keys"]["x"][m[sel]], nb["keys"]["y"][m[sel]], np.ones(len(sel))], 1)
    l = x1 @ F
    d2 = (np.sum(l * x2, 1) ** 2) / (l[:, 0] ** 2 + l[:, 1] ** 2)
    assert np.all(d2 < 3.84 * sg[nb["keys"]["octave"][m[sel]]] * 1.001)
    # coarse skips the epipolar test, only-stereo restricts both sides, the rotation histogram only removes matches
    nc, mc = oracle.search_for_triangulation(cur, nb, cam4, sf, sg, coarse=True)
    assert nc >= n
    ns, ms = oracle.search_for_triangulation(cur, nb, cam4, sf, sg, only_stereo=True)
    s2 = np.nonzero(ms >= 0)[0]
    assert np.all(cur["u_right"][s2] >= 0) and np.all(nb["u_right"][ms[s2]] >= 0)
    no, mo = oracle.search_for_triangulation(cur, nb, cam4, sf, sg, check_orientation=True)
    assert no <= n and np.all((mo == m) | (mo == -1))


def test_oracle_create_new_map_points(oracle, synthetic, keyframes):
    cam4, mbf, mb = cam_of(synthetic)
    sf, sg = tables()
    cur, neigh = keyframes[0], keyframes[1:]
    idx, x3 = oracle.create_new_map_points(cur, neigh, cam4, mb, mbf, sf, sg)
    assert len(idx) > 60
    assert len(np.unique(idx[:, 0])) == len(idx)                      # one point per keypoint of the current keyframe
    assert np.all(np.diff(idx[:, 1]) >= 0)                           # neighbour-major creation order
    for j in np.unique(idx[:, 1]):
        assert np.all(np.diff(idx[idx[:, 1] == j, 0]) > 0)           # keypoint-ascending within a neighbour
    assert not cur["has_point"][idx[:, 0]].any()
    assert (idx[:, 3] == 1).any() and (idx[:, 3] == 0).any()        # stereo un-projections and triangulations both occur
    # the points re-project onto their keypoints in the current keyframe
    Xc = x3 - cur["centre"].astype(np.float32)
    u = cam4[0] * Xc[:, 0] / Xc[:, 2] + cam4[2]
    v = cam4[1] * Xc[:, 1] / Xc[:, 2] + cam4[3]
    err = np.hypot(u - cur["keys"]["x"][idx[:, 0]], v - cur["keys"]["y"][idx[:, 0]])
    assert np.all(Xc[:, 2] > 0) and np.median(err) < 1.5 and err.max() < 3.0 * np.sqrt(sg.max())
    # a neighbour closer than the stereo baseline is skipped (LocalMapping.cc:456-460)
    near = dict(neigh[0]); near["pose7"] = cur["pose7"].copy(); near["pose7"][4] -= 0.5 * mb
    idx2, _ = oracle.create_new_map_points(cur, [near], cam4, mb, mbf, sf, sg)
    assert len(idx2) == 0
    # far-point limit
    idx3, x33 = oracle.create_new_map_points(cur, neigh, cam4, mb, mbf, sf, sg, far_points=True, th_far_points=15.0)
    assert 0 < len(idx3) < len(idx) and np.all(np.linalg.norm(x33 - cur["centre"].astype(np.float32), axis=1) < 15.0)


@pytest.mark.gpu
def test_product_matches_the_oracle(pkg, oracle, synthetic, keyframes):
    cam4, mbf, mb = cam_of(synthetic)
    cam5 = np.float32([cam4[0], cam4[1], cam4[2], cam4[3], mbf]).astype(np.float64)
    sf, sg = tables()
    cur = keyframes[0]
    for nb in keyframes[1:]:
        for kw in (dict(), dict(coarse=True), dict(only_stereo=True), dict(check_orientation=True)):
            want = oracle.search_for_triangulation(cur, nb, cam4, sf, sg, **kw)
            got = pkg.capi.search_for_triangulation(cur, nb, cam5, sf, sg, **kw)
            assert got[0] == want[0] and np.array_equal(got[1], want[1]), kw
    for kw in (dict(), dict(inertial=True), dict(far_points=True, th_far_points=15.0), dict(coarse=True)):
        want = oracle.create_new_map_points(cur, keyframes[1:], cam4, mb, mbf, sf, sg, **kw)
        got = pkg.capi.create_new_map_points(cur, keyframes[1:], cam5, mb, sf, sg, **kw)
        assert np.array_equal(got[0], want[0]), kw
        stereo = want[0][:, 3] == 1
        assert np.array_equal(got[1][stereo], want[1][stereo])                       # un-projections: bit for bit
        assert np.allclose(got[1][~stereo], want[1][~stereo], rtol=1e-4, atol=1e-5)  # triangulations: float rounding of the eigenvector
    # fewer neighbours, an empty neighbour list, a keyframe without vocabulary entries
    got = pkg.capi.create_new_map_points(cur, [], cam5, mb, sf, sg)
    assert len(got[0]) == 0
    bare = dict(cur); bare["fv_node"] = np.zeros(0, np.int32); bare["fv_offset"] = np.zeros(1, np.int32); bare["fv_index"] = np.zeros(0, np.int32)
    assert pkg.capi.search_for_triangulation(bare, keyframes[1], cam5, sf, sg)[0] == 0
    bad = dict(cur); bad["fv_index"] = cur["fv_index"].copy(); bad["fv_index"][0] = 10 ** 6
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.search_for_triangulation(bad, keyframes[1], cam5, sf, sg)


# ---- ORBmatcher::Fuse, the search -----------------------------------------------------------------------------------------------
def fuse_problem(oracle, keyframes, synthetic, src=2, dst=0, seed=0):
    """Map points = the stereo points of keyframe `src`; they are fused into keyframe `dst`."""
    cam4, mbf, mb = cam_of(synthetic)
    sf, sg = tables()
    rng = np.random.default_rng(seed)
    A, B = keyframes[src], keyframes[dst]
    sel = np.nonzero(A["depth"] > 0)[0]
    z = A["depth"][sel]
    Xc = np.stack([(A["keys"]["x"][sel] - cam4[2]) * z / cam4[0], (A["keys"]["y"][sel] - cam4[3]) * z / cam4[1], z], 1).astype(np.float32)
    Xw = (Xc + A["centre"].astype(np.float32)).astype(np.float32)
    pts = np.zeros(len(sel), oracle.MAP_POINT_DTYPE)
    pts["pos"] = Xw
    v = Xw - A["centre"].astype(np.float32)
    dist = np.linalg.norm(v, axis=1).astype(np.float32)
    pts["normal"] = v / dist[:, None]
    raw = (dist * sf[A["keys"]["octave"][sel]]).astype(np.float32)
    pts["max_distance_raw"] = raw
    pts["max_distance"] = np.float32(1.2) * raw
    pts["min_distance"] = np.float32(0.8) * (raw / sf[-1])
    pts["descriptor"] = A["descriptors"][sel]
    valid = (rng.random(len(sel)) < 0.85).astype(np.uint8)
    return B, pts, valid, cam4, mbf, sf, (np.float32(1) / sg).astype(np.float32), float(np.log(np.float32(1.2)))


def test_oracle_fuse_search(oracle, synthetic, keyframes):
    B, pts, valid, cam4, mbf, sf, isg, logsf = fuse_problem(oracle, keyframes, synthetic)
    nf, bi, bd = oracle.fuse_search(B["keys"], B["descriptors"], B["u_right"], W, H, B["pose7"], cam4, mbf, sf, isg, logsf, pts, valid, th=3.0)
    assert nf == (bi >= 0).sum() and nf > 100
    assert not (bi[valid == 0] >= 0).any()
    hit = np.nonzero(bi >= 0)[0]
    assert np.all(bd[hit] <= 50)
    # the fused keypoint lies where the point projects
    Xc = pts["pos"][hit] - B["centre"].astype(np.float32)
    u = cam4[0] * Xc[:, 0] / Xc[:, 2] + cam4[2]
    v = cam4[1] * Xc[:, 1] / Xc[:, 2] + cam4[3]
    err = np.hypot(u - B["keys"]["x"][bi[hit]], v - B["keys"]["y"][bi[hit]])
    assert np.median(err) < 1.5 and np.all(err < 3.0 * sf[B["keys"]["octave"][bi[hit]]] * np.sqrt(2) + 1e-3)
    # a wider window cannot lose candidates that pass the gates; behind the camera nothing is fused
    nf2, bi2, _ = oracle.fuse_search(B["keys"], B["descriptors"], B["u_right"], W, H, B["pose7"], cam4, mbf, sf, isg, logsf, pts, valid, th=6.0)
    assert nf2 >= nf
    back = B["pose7"].copy(); back[6] -= 500.0
    assert oracle.fuse_search(B["keys"], B["descriptors"], B["u_right"], W, H, back, cam4, mbf, sf, isg, logsf, pts, valid)[0] == 0


@pytest.mark.gpu
def test_product_fuse_search(pkg, oracle, synthetic, keyframes):
    for src, dst, th in [(2, 0, 3.0), (0, 3, 3.0), (1, 4, 6.0)]:
        B, pts, valid, cam4, mbf, sf, isg, logsf = fuse_problem(oracle, keyframes, synthetic, src, dst, seed=src)
        want = oracle.fuse_search(B["keys"], B["descriptors"], B["u_right"], W, H, B["pose7"], cam4, mbf, sf, isg, logsf, pts, valid, th=th)
        got = pkg.capi.fuse_search(B["keys"], B["descriptors"], B["u_right"], W, H, B["pose7"], cam4, mbf, sf, isg, logsf, pts, valid, th=th)
        assert got[0] == want[0] and want[0] > 50
        assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    got = pkg.capi.fuse_search(B["keys"], B["descriptors"], B["u_right"], W, H, B["pose7"], cam4, mbf, sf, isg, logsf, pts[:0], valid[:0])
    assert got[0] == 0
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.fuse_search(B["keys"], B["descriptors"], B["u_right"], W, H, B["pose7"], cam4, mbf, sf, isg, 0.0, pts, valid)
