"""CPU tests of the LiDAR oracle against independent numpy statements: preprocess filter, voxel-grid centroids,
exact k nearest neighbours (brute force), plane fit (numpy lstsq) and the feature_extraction gates."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def scans(synthetic):
    sc = synthetic.Scene(2)
    return [synthetic.lidar_scan(sc, f) for f in range(3)]


def xyz(p):
    return np.stack([p["x"], p["y"], p["z"]], 1)


def test_preprocess(oracle, scans):
    raw = scans[0]
    out = oracle.lidar_preprocess(raw, point_filter_num=2, blind=5.0, time_unit_scale=1e-3)
    idx = np.arange(len(raw))
    r2 = raw["x"] * raw["x"] + raw["y"] * raw["y"] + raw["z"] * raw["z"]
    keep = (idx % 2 == 0) & (r2.astype(np.float64) > 25.0)
    assert keep.sum() < (idx % 2 == 0).sum()  # the blind radius removes something
    assert len(out) == keep.sum()
    assert np.array_equal(out["x"], raw["x"][keep]) and np.array_equal(out["intensity"], raw["intensity"][keep])
    assert np.array_equal(out["curvature"], raw["time"][keep] * np.float32(1e-3))
    assert np.all(out["normal_x"] == 0) and np.all(out["pad0"] == 1.0)
    assert len(oracle.lidar_preprocess(raw[:0])) == 0
    assert len(oracle.lidar_preprocess(raw, point_filter_num=3)) == len(raw[::3])


def test_voxel_grid(oracle, scans):
    pts = oracle.lidar_preprocess(scans[0])
    out = oracle.voxel_grid(pts, 0.5)
    inv = np.float32(1.0) / np.float32(0.5)
    P = xyz(pts)
    mn, mx = P.min(0), P.max(0)
    min_b = np.floor(mn * inv).astype(np.int64)
    div_b = np.floor(mx * inv).astype(np.int64) - min_b + 1
    ijk = (np.floor(P * inv) - min_b.astype(np.float32)).astype(np.int64)
    idx = ijk[:, 0] + ijk[:, 1] * div_b[0] + ijk[:, 2] * div_b[0] * div_b[1]
    uniq, inverse, counts = np.unique(idx, return_inverse=True, return_counts=True)
    assert len(out) == len(uniq)
    # centroids in ascending voxel order; float32 sequential sums vs float64 means agree to float accuracy
    for name in ("x", "y", "z", "intensity", "curvature"):
        ref = np.bincount(inverse, weights=pts[name].astype(np.float64)) / counts
        assert np.allclose(out[name], ref, rtol=2e-6, atol=2e-5), name
    # bit-exact check of a few voxels by replaying the float32 running sum in point order
    for v in (0, len(uniq) // 2, len(uniq) - 1):
        members = np.nonzero(inverse == v)[0]
        acc = np.float32(0)
        for m in members:
            acc = np.float32(acc + pts["x"][m])
        assert out["x"][v] == np.float32(acc / np.float32(len(members)))
    assert len(oracle.voxel_grid(pts[:1], 0.5)) == 1


def test_kdtree_exact_knn(oracle):
    rng = np.random.default_rng(0)
    M = np.zeros(4000, oracle.POINT_DTYPE)
    M["x"], M["y"], M["z"] = rng.uniform(-30, 30, 4000), rng.uniform(-30, 30, 4000), rng.uniform(-2, 4, 4000)
    Q = np.zeros(300, oracle.POINT_DTYPE)
    Q["x"], Q["y"], Q["z"] = rng.uniform(-35, 35, 300), rng.uniform(-35, 35, 300), rng.uniform(-3, 5, 300)
    tree = oracle.KdTree(M[:3000])
    tree.add(M[3000:])
    assert tree.size() == 4000
    near, d, found = tree.knn(Q, 5)
    assert np.all(found == 5)
    Mx, Qx = xyz(M), xyz(Q)
    for i in range(len(Q)):
        diff = Qx[i] - Mx
        d2 = (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]  # float32, ikd-Tree's order
        order = np.argsort(d2, kind="stable")[:5]
        assert np.array_equal(d[i], d2[order])
        assert np.array_equal(xyz(near[i]), Mx[order])
    # fewer map points than k
    small = oracle.KdTree(M[:3])
    near, d, found = small.knn(Q[:4], 5)
    assert np.all(found == 3)


def test_esti_plane(oracle):
    rng = np.random.default_rng(1)
    ok_count = 0
    for t in range(200):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        d0 = rng.uniform(2, 30)
        basis = np.linalg.svd(n[None])[2][1:]
        p = (-d0 * n)[None] + rng.uniform(-1, 1, (5, 2)) @ basis + rng.normal(0, 0.01 if t % 4 else 0.2, (5, 1)) * n[None]
        five = np.zeros(5, oracle.POINT_DTYPE)
        five["x"], five["y"], five["z"] = p[:, 0], p[:, 1], p[:, 2]
        ok, pabcd = oracle.esti_plane(five, 0.1)
        A = xyz(five).astype(np.float64)
        sol = np.linalg.lstsq(A, -np.ones(5), rcond=None)[0]
        nn = np.linalg.norm(sol)
        ref = np.concatenate([sol / nn, [1 / nn]])
        assert np.allclose(pabcd, ref, rtol=0, atol=5e-4 * max(1.0, d0 / 5)), (t, pabcd, ref)
        resid = np.abs(A @ ref[:3] + ref[3])
        if resid.max() < 0.09:
            assert ok
        if resid.max() > 0.11:
            assert not ok
        ok_count += ok
    assert 100 < ok_count < 200


def test_feature_extraction(oracle, synthetic, scans):
    d = [oracle.voxel_grid(oracle.lidar_preprocess(s)) for s in scans]
    st = [oracle.pack_state(*synthetic.lidar_state(f)) for f in range(3)]
    # map = world-frame points of scan 0 (its pose is the identity up to the tiny y offset)
    fe0 = oracle.feature_extraction(oracle.KdTree(d[0][:10]), d[0], st[0])
    tree = oracle.KdTree(fe0["world"])
    fe = oracle.feature_extraction(tree, d[1], st[1])
    n = len(d[1])
    assert fe["effct_feat_num"] == fe["selected"].sum() and 0.5 * n < fe["effct_feat_num"] <= n
    R, t = synthetic.sensor_pose(1)
    W = xyz(d[1]).astype(np.float64) @ R.T + t
    assert np.allclose(xyz(fe["world"]), W, atol=1e-4)
    sel = fe["selected"] > 0
    assert np.array_equal(xyz(fe["cloud_ori"]), xyz(d[1])[sel])  # body-frame points, original order
    nv = fe["normvec"][sel]
    assert np.allclose(np.linalg.norm(xyz(nv), axis=1), 1.0, atol=1e-5)
    # selected points lie on their plane: |pd2| small relative to sqrt(range)
    rng_body = np.linalg.norm(xyz(d[1])[sel], axis=1)
    assert np.all(1 - 0.9 * np.abs(nv["intensity"]) / np.sqrt(rng_body) > 0.9 - 1e-6)
    assert np.all(fe["nfound"] == 5)
    # the ground dominates: many normals are close to +-z
    assert (np.abs(nv["z"]) > 0.95).mean() > 0.3


def test_tree_map_operations_equal_the_point_list(oracle, synthetic):
    """The CPU baseline keeps its map in the k-d tree (Add_Points with down-sampling and box deletion as lazily flagged nodes, the way
    ikd-Tree does it, ikd_Tree.cpp:478-584,779-860) -- the same operations on the plain point list are the statement the GPU map
    maintenance is tested against, so the two must leave the same multiset of points."""
    rng = np.random.default_rng(5)
    base = np.zeros(4000, oracle.POINT_DTYPE)
    base["x"], base["y"], base["z"] = rng.uniform(-10, 10, 4000), rng.uniform(-10, 10, 4000), rng.uniform(-1, 1, 4000)
    tree = oracle.KdTree(base)
    lst = base.copy()
    for step in range(4):
        add = np.zeros(1500, oracle.POINT_DTYPE)
        add["x"], add["y"], add["z"] = rng.uniform(-12, 12, 1500), rng.uniform(-12, 12, 1500), rng.uniform(-1, 1, 1500)
        add["intensity"] = step
        oracle.kdtree_add_points(tree, add, downsample=True, size=0.5)
        lst = oracle.mappoints_add(lst, add, downsample=True, size=0.5)
        plain = add[:100].copy(); plain["x"] += 30
        oracle.kdtree_add_points(tree, plain, downsample=False)
        lst = oracle.mappoints_add(lst, plain, downsample=False)
        box = np.array([[-3 + step, -3, -1, 0 + step, 2, 1]], np.float32)
        n_del = oracle.kdtree_delete_boxes(tree, box)
        lst2 = oracle.map_delete_boxes(lst, box)
        assert n_del == len(lst) - len(lst2) and n_del > 0
        lst = lst2
        got = oracle.kdtree_valid_points(tree)
        assert len(got) == len(lst)
        key = lambda a: np.lexsort((a["intensity"], a["z"], a["y"], a["x"]))
        assert np.array_equal(got[key(got)], lst[key(lst)])
    # the search skips the flagged points: same neighbours as a tree built from the surviving points
    q = np.zeros(300, oracle.POINT_DTYPE)
    q["x"], q["y"], q["z"] = rng.uniform(-12, 12, 300), rng.uniform(-12, 12, 300), rng.uniform(-1, 1, 300)
    fresh = oracle.KdTree(lst)
    a, b = tree.knn(q), fresh.knn(q)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])


def test_synthetic_street_map_is_reference_sized(synthetic):
    m = synthetic.lidar_map(synthetic.Scene(0))
    assert 1e5 < len(m) < 1e6
    # one point per 0.5 m cell
    cells = np.floor(np.stack([m["x"], m["y"], m["z"]], 1) / 0.5).astype(np.int64)
    assert len(np.unique(cells, axis=0)) > 0.97 * len(m)
