"""GPU parity of the HIP LiDAR front end with the oracle.  Preprocess output, voxel centroids, world points,
neighbour sets/distances, selection mask, plane normals and residuals are all compared for equality (the
tolerance stated by BASELINE.md -- mask identical, (n, d, residual) within 1e-6 -- is met with zero error)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def same_points(a, b):
    return a.tobytes() == b.tobytes()


@pytest.fixture(scope="module")
def fe(pkg):
    assert pkg.device_count() >= 1
    f = pkg.LidarFrontEnd(max_points_per_scan=140000, max_scans=4)
    yield f
    f.close()


@pytest.fixture(scope="module")
def scans(synthetic):
    sc = synthetic.Scene(2)
    return [synthetic.lidar_scan(sc, f) for f in range(4)]


@pytest.mark.parametrize("form", ["blocks", "stream"])  # the two device forms (lidar_host.cpp run_preprocess): three launches over 1024-point blocks /
@pytest.mark.parametrize("pfn,blind", [(2, 2.0), (1, 5.0), (3, 8.5), (4, 0.0)])  # one pass per scan (k_pre_stream, batches of scans)
def test_preprocess(fe, oracle, scans, pfn, blind, form, monkeypatch):
    monkeypatch.setenv("TC2LI_PRE_STREAM", "1" if form == "stream" else "0")
    raw = scans[0]
    got = fe.process(raw, pfn, blind, 1e-3)
    want = oracle.lidar_preprocess(raw, pfn, blind, 1e-3)
    assert len(got) == len(want) and same_points(got, want)
    assert len(fe.process(raw[:0])) == 0
    assert same_points(fe.process(raw[:1], 2, 0.0), oracle.lidar_preprocess(raw[:1], 2, 0.0))
    assert same_points(fe.process(raw[:1025], 2, 2.0), oracle.lidar_preprocess(raw[:1025], 2, 2.0))


# the device forms of the filter (lidar_host.cpp run_voxel): few scans / batches; the batches' sums in one kernel or in two passes
@pytest.mark.parametrize("form", ["hash", "sorted", "sorted-two-pass"])
@pytest.mark.parametrize("leaf", [0.5, 0.2, 1.5])
def test_voxel_filter(fe, oracle, scans, leaf, form, monkeypatch):
    monkeypatch.setenv("TC2LI_VOXEL_SORTED", "0" if form == "hash" else "1")
    monkeypatch.setenv("TC2LI_VOXEL_FUSED", "0" if form == "sorted-two-pass" else "1")
    pts = oracle.lidar_preprocess(scans[1])
    got = fe.voxel_filter(pts, leaf)
    want = oracle.voxel_grid(pts, leaf)
    assert len(got) == len(want)
    for name in ("x", "y", "z", "intensity", "curvature", "normal_x"):
        assert np.array_equal(got[name], want[name]), name
    assert same_points(got, want)
    assert same_points(fe.voxel_filter(pts[:1], leaf), oracle.voxel_grid(pts[:1], leaf))
    # many points in one voxel (long sequential sums)
    dense = pts[:3000].copy()
    dense["x"] = dense["x"] * np.float32(0.01); dense["y"] *= np.float32(0.01); dense["z"] *= np.float32(0.01)
    assert same_points(fe.voxel_filter(dense, leaf), oracle.voxel_grid(dense, leaf))
    # one voxel with more members than the LDS stage of the centroid kernel (4096), next to ordinary ones
    huge = pts[:12000].copy()
    for name in ("x", "y", "z"):
        huge[name][:9000] = huge[name][:9000] * np.float32(0.002) + np.float32(0.1)
    assert same_points(fe.voxel_filter(huge, leaf), oracle.voxel_grid(huge, leaf))
    # points that are not finite are left out; coordinates on both sides of zero and a wide box (three or four key digits)
    wide = pts[:20000].copy()
    wide["x"][::7] *= np.float32(-3.0); wide["y"][::5] *= np.float32(4.0)
    wide["x"][11::97] = np.float32(np.nan); wide["z"][5::131] = np.float32(np.inf)
    assert same_points(fe.voxel_filter(wide, leaf), oracle.voxel_grid(wide, leaf))
    assert len(fe.voxel_filter(pts[:0], leaf)) == 0


def test_frontend_batch_preprocess_and_filter_forms(pkg, oracle, scans, monkeypatch):
    """tc2li_lidar_frontend_batch with the one-pass preprocess (k_pre_stream, which also hands the voxel filter its bounding boxes) and with
    the three-launch form: the oracle's numbers of preprocessed and down-sampled points per scan, and the same selected features byte for
    byte from both forms -- ragged scans, an empty one, NaN / inf coordinates."""
    import torch
    batch = [scans[0], scans[1][:50001].copy(), scans[2][:0], scans[3][:777]]
    batch[1]["x"][10::101] = np.float32(np.nan); batch[1]["z"][6::313] = np.float32(np.inf)
    f4 = pkg.LidarFrontEnd(max_points_per_scan=140000, max_scans=4)
    raw = torch.from_numpy(np.concatenate(batch).view(np.uint8)).cuda()
    offs = np.concatenate([[0], np.cumsum([len(b) for b in batch])]).astype(np.int32)
    boot = oracle.voxel_grid(oracle.lidar_preprocess(scans[0]))
    maps = []
    for _ in batch:
        m = pkg.LidarMap(); m.Build(boot); maps.append(m)
    st = np.stack([pkg.pack_lidar_state(np.eye(3), np.zeros(3))] * 4)
    out = {}
    for stream in (0, 1):
        monkeypatch.setenv("TC2LI_PRE_STREAM", str(stream))
        counts, ori, corr = f4.frontend_batch(raw.data_ptr(), offs, maps, st, want_points=True)
        for s, b in enumerate(batch):
            pre = oracle.lidar_preprocess(b)
            assert int(counts[0][s]) == len(pre) and int(counts[1][s]) == len(oracle.voxel_grid(pre)), (stream, s)
        out[stream] = (counts.copy(), ori.copy(), corr.copy())
    assert out[0][0][2][0] > 1000  # the first scan finds its features in the map of its own points
    for a, b in zip(out[0], out[1]):
        assert a.tobytes() == b.tobytes()
    f4.close()


def test_frontend_batch_voxel_coordinates_from_the_preprocess_pass(pkg, oracle, scans, monkeypatch):
    """Round 5: in a batch of >= 32 scans k_pre_stream leaves the voxel filter the kept points' packed voxel coordinates (the filter's sort then
    does not open the 48-byte records again).  40 ragged scans -- one with NaN / inf coordinates and one reaching beyond +-1024 leaves, for
    which the filter reads the points itself -- give the oracle's counts and the same features byte for byte with and without the packed
    coordinates (TC2LI_VOXEL_PRE_KEYS=0)."""
    import torch
    batch = [scans[k % 4][:(30000 + 2500 * k)].copy() for k in range(40)]
    batch[3]["x"][10::101] = np.float32(np.nan); batch[3]["z"][6::313] = np.float32(np.inf)
    batch[7]["x"][5::997] += np.float32(600.0)   # floor(x / 0.5) beyond 1023: this scan's coordinates do not fit the packing
    batch[11] = batch[11][:0]
    f = pkg.LidarFrontEnd(max_points_per_scan=140000, max_scans=40)
    raw = torch.from_numpy(np.concatenate(batch).view(np.uint8)).cuda()
    offs = np.concatenate([[0], np.cumsum([len(b) for b in batch])]).astype(np.int32)
    boot = oracle.voxel_grid(oracle.lidar_preprocess(scans[0]))
    maps = []
    for _ in batch:
        m = pkg.LidarMap(); m.Build(boot); maps.append(m)
    st = np.stack([pkg.pack_lidar_state(np.eye(3), np.zeros(3))] * len(batch))
    monkeypatch.setenv("TC2LI_PRE_STREAM", "1")
    out = {}
    for keys in (1, 0):
        monkeypatch.setenv("TC2LI_VOXEL_PRE_KEYS", str(keys))
        counts, ori, corr = f.frontend_batch(raw.data_ptr(), offs, maps, st, want_points=True)
        out[keys] = (counts.copy(), ori.copy(), corr.copy())
    for s in (0, 3, 7, 11, 39):
        pre = oracle.lidar_preprocess(batch[s])
        assert int(out[1][0][0][s]) == len(pre) and int(out[1][0][1][s]) == len(oracle.voxel_grid(pre)), s
    for a, b in zip(out[0], out[1]):
        assert a.tobytes() == b.tobytes()
    f.close()


def test_feature_extraction(pkg, fe, oracle, synthetic, scans):
    down = [oracle.voxel_grid(oracle.lidar_preprocess(s)) for s in scans[:3]]
    st = [oracle.pack_state(*synthetic.lidar_state(f)) for f in range(3)]
    boot = oracle.feature_extraction(oracle.KdTree(down[0][:10]), down[0], st[0])
    world0 = boot["world"]
    tree = oracle.KdTree(world0)
    gmap = pkg.LidarMap()
    assert gmap.Build(world0) == len(world0)
    for f in (1, 2):
        want = oracle.feature_extraction(tree, down[f], st[f])
        got = fe.feature_extraction(gmap, down[f], st[f])
        assert same_points(got["world"], want["world"])
        assert np.array_equal(got["nfound"], want["nfound"])
        assert np.array_equal(got["sqdist"], oracle.KdTree.knn(tree, want["world"])[1])
        assert same_points(got["nearest"], want["nearest"])
        assert np.array_equal(got["selected"], want["selected"])
        sel = want["selected"] > 0
        for name in ("x", "y", "z", "intensity"):
            assert np.array_equal(got["normvec"][name][sel], want["normvec"][name][sel]), name
        assert got["effct_feat_num"] == want["effct_feat_num"] > 1000
        assert same_points(got["cloud_ori"], want["cloud_ori"])
        assert same_points(got["corr_normvect"], want["corr_normvect"])
        # grow the map like map_incremental's plain insertions do
        tree.add(want["world"])
        gmap.Add_Points(want["world"])
    assert gmap.size() == tree.size()
    gmap.close()


def test_knn_edge_cases(pkg, fe, oracle):
    rng = np.random.default_rng(5)
    M = np.zeros(2000, oracle.POINT_DTYPE)
    M["x"], M["y"], M["z"] = rng.uniform(-20, 20, 2000), rng.uniform(-20, 20, 2000), rng.uniform(-2, 2, 2000)
    Q = np.zeros(500, oracle.POINT_DTYPE)
    Q["x"], Q["y"], Q["z"] = rng.uniform(-60, 60, 500), rng.uniform(-60, 60, 500), rng.uniform(-10, 10, 500)  # many far queries
    st = oracle.pack_state(np.eye(3), np.zeros(3), np.eye(3), np.zeros(3))
    for nmap in (2000, 7, 3, 0):
        tree = oracle.KdTree(M[:nmap])
        gmap = pkg.LidarMap()
        gmap.Build(M[:nmap])
        want = oracle.feature_extraction(tree, Q, st)
        got = fe.feature_extraction(gmap, Q, st)
        assert np.array_equal(got["nfound"], want["nfound"])
        assert same_points(got["nearest"], want["nearest"])
        assert np.array_equal(got["selected"], want["selected"])
        gmap.close()
    # duplicated map points: equal distances are ordered by x like PointType_CMP
    D = np.concatenate([M[:50], M[:50]])
    D["x"][50:] += np.float32(0.0)
    tree, gmap = oracle.KdTree(D), pkg.LidarMap()
    gmap.Build(D)
    want = oracle.feature_extraction(tree, Q[:100], st)
    got = fe.feature_extraction(gmap, Q[:100], st)
    assert np.array_equal(got["sqdist"], oracle.KdTree.knn(tree, want["world"])[1])
    gmap.close()


def test_frontend_batch(pkg, fe, oracle, synthetic, scans):
    import torch
    down0 = oracle.voxel_grid(oracle.lidar_preprocess(scans[0]))
    st0 = oracle.pack_state(*synthetic.lidar_state(0))
    world0 = oracle.feature_extraction(oracle.KdTree(down0[:10]), down0, st0)["world"]
    tree = oracle.KdTree(world0)
    maps = [pkg.LidarMap() for _ in range(3)]
    for m in maps:
        m.Build(world0)
    batch = [scans[1], scans[2], scans[3][:50000]]
    offs = np.concatenate([[0], np.cumsum([len(b) for b in batch])]).astype(np.int32)
    raw = np.concatenate(batch)
    dev = torch.from_numpy(raw.view(np.uint8)).cuda()
    states = np.stack([oracle.pack_state(*synthetic.lidar_state(f)) for f in (1, 2, 3)])
    counts, ori, corr = fe.frontend_batch(dev.data_ptr(), offs, maps, states)
    for s, f in enumerate((1, 2, 3)):
        pre = oracle.lidar_preprocess(batch[s])
        down = oracle.voxel_grid(pre)
        want = oracle.feature_extraction(tree, down, states[s])
        assert counts[0, s] == len(pre) and counts[1, s] == len(down) and counts[2, s] == want["effct_feat_num"]
        m = want["effct_feat_num"]
        assert same_points(ori[s, :m], want["cloud_ori"])
        assert same_points(corr[s, :m], want["corr_normvect"])
    for m in maps:
        m.close()
