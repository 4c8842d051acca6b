"""GPU parity of the persistent map maintenance (map_incremental, Add_Points with
down-sampling, Delete_Point_Boxes) with the oracle.  The map is a multiset of points: compared after sorting."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def canon(p):
    a = np.stack([p["x"], p["y"], p["z"], p["intensity"], p["curvature"]], 1)
    return a[np.lexsort(a.T[::-1])]


@pytest.fixture(scope="module")
def setup(pkg, oracle, synthetic):
    fe = pkg.LidarFrontEnd(max_points_per_scan=140000, max_scans=1)
    scene = synthetic.Scene(0)
    downs, states = [], []
    for f in range(4):
        downs.append(oracle.voxel_grid(oracle.lidar_preprocess(synthetic.lidar_scan(scene, f))))
        states.append(pkg.pack_lidar_state(*synthetic.lidar_state(f)[:2]))
    boot = pkg.LidarMap()
    boot.Build(downs[0][:8])
    world0 = fe.feature_extraction(boot, downs[0], states[0])["world"]
    return fe, downs, states, world0


@pytest.mark.parametrize("ekf,fs", [(True, 0.5), (False, 0.5), (True, 0.8)])
def test_map_incremental_sequence(pkg, oracle, setup, ekf, fs):
    """Three consecutive scans inserted into the map of the first: after every step the device map equals the oracle's."""
    fe, downs, states, world0 = setup
    m = pkg.LidarMap()
    m.Build(world0)
    ref = world0.copy()
    for f in (1, 2, 3):
        # the pose used for the insertion differs slightly from the one of the extraction (UpdateLidarPose in between)
        upd = states[f].copy(); upd[9:12] += [0.02, -0.01, 0.005]
        fe.feature_extraction(m, downs[f], states[f])
        n, na, nn = m.map_incremental(fe, 0, upd, ekf_inited=ekf, filter_size_map_min=fs)
        ref, wa, wn = oracle.map_incremental(ref, downs[f], states[f], upd, ekf_inited=ekf, filter_size_map_min=fs)
        assert (na, nn) == (wa, wn) and n == len(ref)
        got = m.points()
        assert np.array_equal(canon(got), canon(ref))
        assert na > 50 and (nn > 0 or not ekf)
    assert m.size() > len(world0) or fs > 0.5  # a coarser voxel merges stored points


def _grid_is_sound(m):
    """The grid walked on the host (tc2li_lidar_map_grid_download checks every live entry against the cell its coordinates name, the rows'
    room and the tombstones): every point of the map is there exactly once."""
    cells, idx = m.grid()
    assert np.array_equal(np.sort(idx), np.arange(m.size()))
    return cells, idx


def test_in_place_grid_update_equals_rebuild(pkg, oracle, setup, monkeypatch):
    """map_incremental merges the added points into the rows of the existing grid (KD_TREE::Add_Points inserts into the
    existing tree, ikd_Tree.cpp:478-584) instead of rebuilding it.  Two maps take the same twelve insertions, one in place, one with
    TC2LI_MAP_ALWAYS_REBUILD=1 (rounds 1-3): after every step the point lists are equal element for element, both grids are sound cell
    by cell, a search of either map gives the same neighbours / distances / planes, and both equal the oracle's map."""
    fe, downs, states, world0 = setup
    a, b = pkg.LidarMap(), pkg.LidarMap()
    a.Build(world0); b.Build(world0)
    builds0 = a.stats()["grid_builds"]
    ref = world0.copy()
    rng = np.random.default_rng(5)
    for step in range(12):
        f = 1 + step % 3
        upd = states[f].copy(); upd[9:12] += rng.normal(0, 0.15, 3)  # the scans land a little elsewhere every time: new voxels, replaced points
        monkeypatch.delenv("TC2LI_MAP_ALWAYS_REBUILD", raising=False)
        ra = fe.feature_extraction(a, downs[f], states[f])
        na = a.map_incremental(fe, 0, upd, ekf_inited=True, filter_size_map_min=0.5)
        monkeypatch.setenv("TC2LI_MAP_ALWAYS_REBUILD", "1")
        rb = fe.feature_extraction(b, downs[f], states[f])
        nb = b.map_incremental(fe, 0, upd, ekf_inited=True, filter_size_map_min=0.5)
        assert na == nb
        for k in ("selected", "sqdist", "nearest", "normvec"):
            if k in ra:
                assert np.array_equal(ra[k], rb[k]), (step, k)
        ref, wa, wn = oracle.map_incremental(ref, downs[f], states[f], upd, ekf_inited=True, filter_size_map_min=0.5)
        pa, pb = a.points(), b.points()
        assert np.array_equal(pa, pb) and np.array_equal(canon(pa), canon(ref)) and (na[1], na[2]) == (wa, wn), step
        _grid_is_sound(a); _grid_is_sound(b)
    sa, sb = a.stats(), b.stats()
    assert sb["grid_builds"] == builds0 + 12 and sb["grid_updates"] == 0
    # a rebuild only where the grid said it had to (a segment of this small, sparse map has room for 8 more points than it holds; the scans'
    # first visits to new ground overflow some) -- or on top of an update, when the tombstones left elsewhere have piled up
    assert sa["grid_updates"] >= 6 and sa["grid_builds"] + sa["grid_updates"] >= builds0 + 12, sa


def test_compaction_from_the_deletion_list_and_its_fallback(pkg, oracle, setup, monkeypatch):
    """Round 5: a map whose grid is maintained in place compacts from the LIST of the points a step deletes (k_map_compact_list, one workgroup
    per map) instead of three passes over all of its points; a step with more deletions than the list holds (8192) goes through the flag
    passes after all.  Both against the flag passes alone (TC2LI_MAP_COMPACT_LIST=0: the same point array element for element -- the holes
    take the same fillers) and against the oracle, on an ordinary map and on one with twelve points per voxel, of which a scan replaces more
    than the list holds."""
    fe, downs, states, world0 = setup
    rng = np.random.default_rng(11)
    dense = np.repeat(world0, 12)
    for k in ("x", "y", "z"):
        dense[k] = dense[k] + rng.uniform(-0.04, 0.04, len(dense)).astype(np.float32)
    for base, min_deleted in ((world0, 1), (dense, 8193)):
        a, b = pkg.LidarMap(), pkg.LidarMap()
        a.Build(base); b.Build(base)
        ref = base.copy()
        for f in (1, 2):
            upd = states[f].copy(); upd[9:12] += [0.03, -0.02, 0.01]
            monkeypatch.delenv("TC2LI_MAP_COMPACT_LIST", raising=False)
            fe.feature_extraction(a, downs[f], states[f])
            n0 = a.size()
            na = a.map_incremental(fe, 0, upd, ekf_inited=True, filter_size_map_min=0.5)
            monkeypatch.setenv("TC2LI_MAP_COMPACT_LIST", "0")
            fe.feature_extraction(b, downs[f], states[f])
            nb = b.map_incremental(fe, 0, upd, ekf_inited=True, filter_size_map_min=0.5)
            ref, wa, wn = oracle.map_incremental(ref, downs[f], states[f], upd, ekf_inited=True, filter_size_map_min=0.5)
            assert na == nb and (na[1], na[2]) == (wa, wn)
            pa, pb = a.points(), b.points()
            assert np.array_equal(pa, pb) and np.array_equal(canon(pa), canon(ref)), f
            _grid_is_sound(a); _grid_is_sound(b)
            if f == 1:
                assert n0 + na[1] + na[2] - a.size() >= min_deleted  # what the step deleted (the dense map: more than the list holds)


def test_in_place_grid_update_falls_back_to_a_rebuild(pkg, oracle, setup):
    """What the in-place insertion cannot take ends in a rebuild, with the oracle's map either way: a point outside the grid's box (beyond its
    margin of 8 cells), a segment (16 cells of a row) that receives more points than it has room for, a segment with more entries than the
    merge holds."""
    fe, downs, states, world0 = setup
    ident = pkg.pack_lidar_state(np.eye(3), np.zeros(3))
    def P(xyz):
        a = np.zeros(len(xyz), pkg.capi.POINT_DTYPE); a["x"], a["y"], a["z"] = np.array(xyz, np.float32).T; a["pad0"] = 1; return a
    base = P([[10.26, 10.24, 10.25], [20.1, 20.1, 20.1], [30.05, 30.05, 30.05], [30.45, 30.4, 30.45], [30.3, 30.2, 30.2],
              [40.2, 40.2, 40.2], [41, 41, 41], [42, 42, 42], [43, 43, 43], [44, 44, 44], [45, 45, 45]])
    inside = P([[5.3, 5.3, 9.3], [10.1, 10.1, 10.1], [47.3, 46.2, 45.1]])  # within the box's margin (8 cells along x and y, 2 along z)
    outside = P([[5.3, 5.3, 5.3], [-40.2, 10.1, 10.1]])
    row = P(np.stack([np.linspace(10.3, 44.8, 70), np.full(70, 12.3), np.full(70, 12.3)], 1))          # 70 new voxels along one row of the grid: ~28 per segment, room for 4
    long_base = P(np.stack([1.0 + 6.0 * np.arange(2100) / 2100.0, np.full(2100, 12.3), np.full(2100, 12.3)], 1))  # 2100 entries in one 16-cell segment
    one = P([[3.26, 12.8, 12.8]])
    for base, scan, in_place in ((base, inside, True), (base, outside, False), (base, row, False), (long_base, one, False)):
        m = pkg.LidarMap(); m.Build(base)
        builds0 = m.stats()["grid_builds"]
        fe.feature_extraction(m, scan, ident)
        n, na, nn = m.map_incremental(fe, 0, ident, ekf_inited=False)
        want, wa, wn = oracle.map_incremental(base, scan, ident, ident, ekf_inited=False)
        assert (na, nn) == (wa, wn) and n == len(want) and np.array_equal(canon(m.points()), canon(want))
        _grid_is_sound(m)
        st = m.stats()
        assert (st["grid_updates"], st["grid_builds"] - builds0) == ((1, 0) if in_place else (0, 1)), (st, len(scan))
        # and the map stays searchable: the same answers as a map built afresh from its points
        fresh = pkg.LidarMap(); fresh.Build(m.points())
        x, y = fe.feature_extraction(m, downs[1][:2000], ident), fe.feature_extraction(fresh, downs[1][:2000], ident)
        assert np.array_equal(x["selected"], y["selected"]) and np.array_equal(x["sqdist"], y["sqdist"])


def test_insertion_rule_inside_one_voxel(pkg, oracle, setup):
    """Hand-made cases: empty voxel, one stored point closer / farther than the candidate, several stored points, two
    candidates for one voxel (the second sees the first)."""
    fe, downs, states, world0 = setup
    ident = pkg.pack_lidar_state(np.eye(3), np.zeros(3))
    def P(xyz):
        a = np.zeros(len(xyz), pkg.capi.POINT_DTYPE); a["x"], a["y"], a["z"] = np.array(xyz, np.float32).T; a["pad0"] = 1; return a
    base = P([[10.26, 10.24, 10.25], [20.1, 20.1, 20.1], [30.05, 30.05, 30.05], [30.45, 30.4, 30.45], [30.3, 30.2, 30.2],
              [40.2, 40.2, 40.2], [41, 41, 41], [42, 42, 42], [43, 43, 43], [44, 44, 44], [45, 45, 45]])
    scan = P([[5.3, 5.3, 5.3], [10.1, 10.1, 10.1], [20.24, 20.26, 20.25], [30.26, 30.24, 30.26], [40.4, 40.45, 40.4], [40.27, 40.25, 40.26],
              [40.1, 40.1, 40.12], [5.26, 5.25, 5.25]])
    for ekf in (False, True):
        m = pkg.LidarMap(); m.Build(base)
        fe.feature_extraction(m, scan, ident)
        n, na, nn = m.map_incremental(fe, 0, ident, ekf_inited=ekf)
        want, wa, wn = oracle.map_incremental(base, scan, ident, ident, ekf_inited=ekf)
        assert (na, nn) == (wa, wn) and n == len(want)
        assert np.array_equal(canon(m.points()), canon(want))


def test_overflowing_insertion_list_changes_nothing(pkg, oracle, setup):
    """More down-sampled points to insert than one call takes (8192): TC2LI_ERR_CAPACITY, the map as it was (the compaction kernels see
    the batch's overflow word), and the next call on the same map works -- its deletion marks were taken back."""
    fe, downs, states, world0 = setup
    ident = pkg.pack_lidar_state(np.eye(3), np.zeros(3))
    def P(xyz):
        a = np.zeros(len(xyz), pkg.capi.POINT_DTYPE); a["x"], a["y"], a["z"] = np.array(xyz, np.float32).T; a["pad0"] = 1; return a
    base = P([[10.26, 10.24, 10.25], [20.1, 20.1, 20.1], [30.05, 30.05, 30.05], [30.45, 30.4, 30.45], [30.3, 30.2, 30.2], [40.2, 40.2, 40.2]])
    gx, gy = np.meshgrid(np.arange(100), np.arange(95))
    many = P(np.stack([100.3 + gx.ravel(), 200.3 + gy.ravel(), np.full(gx.size, 3.3)], 1))  # 9500 points, one per empty 0.5 m voxel
    m = pkg.LidarMap(); m.Build(base)
    fe.feature_extraction(m, many, ident)
    with pytest.raises(pkg.capi.Tc2liError) as err:
        m.map_incremental(fe, 0, ident, ekf_inited=False)  # before the filter is initialised every point is a PointToAdd
    assert err.value.code == -5 and m.size() == len(base) and np.array_equal(canon(m.points()), canon(base))
    scan = P([[5.3, 5.3, 5.3], [10.1, 10.1, 10.1], [20.24, 20.26, 20.25], [30.26, 30.24, 30.26], [40.4, 40.45, 40.4], [40.27, 40.25, 40.26]])
    fe.feature_extraction(m, scan, ident)
    n, na, nn = m.map_incremental(fe, 0, ident, ekf_inited=True)
    want, wa, wn = oracle.map_incremental(base, scan, ident, ident, ekf_inited=True)
    assert (na, nn) == (wa, wn) and n == len(want) and np.array_equal(canon(m.points()), canon(want))


def test_delete_point_boxes_and_fov_segment(pkg, oracle, setup):
    fe, downs, states, world0 = setup
    m = pkg.LidarMap(); m.Build(world0)
    boxes = np.array([[-5, -50, -5, 5, 50, 5], [20, -10, -3, 40, 10, 10], [1000, 1000, 1000, 1001, 1001, 1001]], np.float32)
    removed = m.Delete_Point_Boxes(boxes)
    want = oracle.map_delete_boxes(world0, boxes)
    assert removed == len(world0) - len(want) and removed > 100
    assert np.array_equal(canon(m.points()), canon(want))
    # searching the edited map still agrees with a fresh one
    fresh = pkg.LidarMap(); fresh.Build(m.points())
    a = fe.feature_extraction(m, downs[1], states[1]); b = fe.feature_extraction(fresh, downs[1], states[1])
    assert np.array_equal(a["selected"], b["selected"]) and np.array_equal(a["sqdist"], b["sqdist"])
    assert m.Delete_Point_Boxes(np.zeros((0, 6), np.float32)) == 0
    # lasermap_fov_segment: the local-map cube follows the sensor
    lm = pkg.capi.LocalMapBox()
    lm7 = np.zeros(7, np.float32)
    for pos in ([0, 0, 0], [10, 0, 0], [49, 5, 0], [52, 60, -1], [300, 300, 40], [301, 300, 40]):
        got = pkg.capi.lidar_fov_segment(lm, pos, cube_len=200.0, det_range=100.0 / 3)
        lm7, want_boxes = oracle.fov_segment(lm7, pos, 200.0, 100.0 / 3)
        assert np.array_equal(got, want_boxes)
        assert np.array_equal(np.array(lm.vertex_min), lm7[:3]) and np.array_equal(np.array(lm.vertex_max), lm7[3:6])


def test_delete_point_boxes_batch_equals_one_map_at_a_time(pkg, oracle, setup):
    """tc2li_lidar_map_delete_boxes_batch: several maps, each with its own boxes (one without boxes, one whose boxes hit nothing, an empty
    map), end as the per-map call and the oracle leave them -- point order included -- and stay searchable."""
    fe, downs, states, world0 = setup
    parts = [world0, world0[::2].copy(), world0[1::3].copy(), world0[:0].copy(), world0[::5].copy()]
    boxes = [np.array([[-5, -50, -5, 5, 50, 5], [20, -10, -3, 40, 10, 10]], np.float32), np.zeros((0, 6), np.float32),
             np.array([[1000, 1000, 1000, 1001, 1001, 1001]], np.float32), np.array([[-5, -50, -5, 5, 50, 5]], np.float32),
             np.array([[-100, -100, -100, 100, 0, 100]], np.float32)]
    maps = []
    for p in parts:
        m = pkg.LidarMap()
        if len(p):
            m.Build(p)
        maps.append(m)
    removed = pkg.capi.delete_point_boxes_batch(maps, boxes)
    for i, (p, b) in enumerate(zip(parts, boxes)):
        want = oracle.map_delete_boxes(p, b) if len(p) and len(b) else p
        assert removed[i] == len(p) - len(want), i
        got = maps[i].points()
        assert np.array_equal(canon(got), canon(want)), i  # the map is a set: holes are filled from the tail, the order is the library's
        if len(p):
            single = pkg.LidarMap(); single.Build(p)
            assert single.Delete_Point_Boxes(b) == removed[i]
            assert np.array_equal(single.points(), got)  # and it is deterministic
    assert removed[0] > 100 and removed[2] == 0 and removed[4] > 0
    fresh = pkg.LidarMap(); fresh.Build(maps[4].points())
    a = fe.feature_extraction(maps[4], downs[1], states[1]); b = fe.feature_extraction(fresh, downs[1], states[1])
    assert np.array_equal(a["selected"], b["selected"]) and np.array_equal(a["sqdist"], b["sqdist"])
    with pytest.raises(Exception):
        pkg.capi.delete_point_boxes_batch([maps[0], maps[0]], [boxes[0], boxes[0]])


def test_map_incremental_batch_equals_one_map_at_a_time(pkg, oracle, synthetic):
    """tc2li_lidar_map_incremental_batch (one launch per phase for all maps) leaves every map exactly as the per-map call and as the
    oracle do; the batch holds maps of different sizes, a scan slot that is not part of the batch and an empty scan."""
    S = 4
    fe = pkg.LidarFrontEnd(max_points_per_scan=140000, max_scans=S)
    one = pkg.LidarFrontEnd(max_points_per_scan=140000, max_scans=1)
    import torch
    scene = synthetic.Scene(2)
    raws = [synthetic.lidar_scan(scene, f) for f in range(1, S + 1)]
    raws[2] = raws[2][:0]  # an empty scan: its map must stay untouched
    states = np.stack([pkg.pack_lidar_state(*synthetic.lidar_state(f)[:2]) for f in range(1, S + 1)])
    street = synthetic.lidar_map(scene, x_from=-60.0, x_to=90.0)
    inits = [street, street[: len(street) // 3], street[::2], street[5::7]]
    maps = [pkg.LidarMap() for _ in range(S)]
    for m, p in zip(maps, inits):
        m.Build(p)
    raw = np.concatenate(raws)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in raws])]).astype(np.int32)
    dev = torch.from_numpy(raw.view(np.uint8)).cuda()
    counts, _, _ = fe.frontend_batch(dev.data_ptr(), offs, maps, states, want_points=False)
    upd = states.copy(); upd[:, 9:12] += [0.02, -0.01, 0.005]
    sel = [0, 2, 3]  # slot 1 is left out of the batch
    na, nn, sz = pkg.capi.map_incremental_batch(fe, sel, [maps[s] for s in sel], upd[sel])
    for k, s in enumerate(sel):
        down = oracle.voxel_grid(oracle.lidar_preprocess(raws[s])) if len(raws[s]) else np.zeros(0, oracle.POINT_DTYPE)
        assert counts[1][s] == len(down)
        if len(down) == 0:
            assert (na[k], nn[k], sz[k]) == (0, 0, len(inits[s])) and np.array_equal(canon(maps[s].points()), canon(inits[s]))
            continue
        want, wa, wn = oracle.map_incremental(inits[s], down, states[s], upd[s])
        assert (na[k], nn[k], sz[k]) == (wa, wn, len(want)) and wa > 50
        assert np.array_equal(canon(maps[s].points()), canon(want))
        # ... and the per-map entry point gives the same
        m1 = pkg.LidarMap(); m1.Build(inits[s])
        one.feature_extraction(m1, down, states[s])
        n1, a1, b1 = m1.map_incremental(one, 0, upd[s])
        assert (n1, a1, b1) == (sz[k], na[k], nn[k]) and np.array_equal(canon(m1.points()), canon(maps[s].points()))
    assert np.array_equal(canon(maps[1].points()), canon(inits[1]))
    # a second frame against the grown maps: the rebuilt grids answer like fresh ones
    counts2, _, _ = fe.frontend_batch(dev.data_ptr(), offs, maps, states, want_points=False)
    fresh = [pkg.LidarMap() for _ in range(S)]
    for f, m in zip(fresh, maps):
        f.Build(m.points())
    counts3, _, _ = fe.frontend_batch(dev.data_ptr(), offs, fresh, states, want_points=False)
    assert np.array_equal(counts2, counts3)
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.map_incremental_batch(fe, [0, 1], [maps[0], maps[0]], upd[:2])  # a map twice in one batch


def test_an_empty_map_beside_maps_on_the_list_path(pkg, oracle, synthetic):
    """a map that was never built (no points, no grid: not lean) in a batch whose other maps all compact from their deletion
    lists.  No block of the flag passes is launched for it (its map has no points), yet k_map_keep_scan must still initialise its kept count
    and bounding box: the slot of the task list carries the words a previous call left there (here: a 10^5-point map's).  The empty map
    ends as the oracle's map_incremental on an empty map, the others as before."""
    import torch
    S = 3
    fe = pkg.LidarFrontEnd(max_points_per_scan=140000, max_scans=S)
    scene = synthetic.Scene(2)
    raws = [synthetic.lidar_scan(scene, f) for f in range(1, S + 1)]
    states = np.stack([pkg.pack_lidar_state(*synthetic.lidar_state(f)[:2]) for f in range(1, S + 1)])
    street = synthetic.lidar_map(scene, x_from=-60.0, x_to=90.0)
    raw = np.concatenate(raws)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in raws])]).astype(np.int32)
    dev = torch.from_numpy(raw.view(np.uint8)).cuda()
    # first call: three built maps, so that task slot 2 is left holding a large map's kept count
    maps = [pkg.LidarMap() for _ in range(S)]
    for m in maps:
        m.Build(street)
    fe.frontend_batch(dev.data_ptr(), offs, maps, states, want_points=False)
    pkg.capi.map_incremental_batch(fe, [0, 1, 2], maps, states)
    # second call: slots 0 and 1 take the list path (their grids are maintained in place), slot 2 is a map without points
    inits = [maps[0].points(), maps[1].points(), np.zeros(0, pkg.capi.POINT_DTYPE)]
    maps[2] = pkg.LidarMap()
    counts, _, _ = fe.frontend_batch(dev.data_ptr(), offs, maps, states, want_points=False)
    na, nn, sz = pkg.capi.map_incremental_batch(fe, [0, 1, 2], maps, states)
    for s in range(S):
        down = oracle.voxel_grid(oracle.lidar_preprocess(raws[s]))
        want, wa, wn = oracle.map_incremental(inits[s], down, states[s], states[s])
        assert (na[s], nn[s], sz[s]) == (wa, wn, len(want)), s
        assert np.array_equal(canon(maps[s].points()), canon(want)), s
    assert 1000 < sz[2] <= counts[1][2]            # the scan's own points (one per map voxel: the insertion rule down-samples among them)
    assert maps[0].stats()["grid_updates"] >= 1    # ... and the other two did take the list path


def test_reference_sized_map(pkg, oracle, synthetic):
    """the reference's map holds 10^5 - 10^6 points.  Feature extraction and map_incremental against the
    ~1.9 * 10^5-point street map bench.py uses: identical neighbours, selection and inserted points as the oracle's k-d tree."""
    scene = synthetic.Scene(1)
    street = synthetic.lidar_map(scene)
    assert len(street) > 150000
    fe = pkg.LidarFrontEnd(max_points_per_scan=140000, max_scans=1)
    m = pkg.LidarMap(); m.Build(street)
    tree = oracle.KdTree(street)
    ref = street
    for f in (3, 4):
        down = oracle.voxel_grid(oracle.lidar_preprocess(synthetic.lidar_scan(scene, f)))
        st = pkg.pack_lidar_state(*synthetic.lidar_state(f)[:2])
        got = fe.feature_extraction(m, down, st)
        want = oracle.feature_extraction(tree, down, st)
        assert np.array_equal(got["selected"], want["selected"]) and got["selected"].sum() > 2000
        assert np.array_equal(got["nfound"], want["nfound"])
        assert np.array_equal(got["nearest"], want["nearest"])
        assert np.array_equal(got["normvec"][got["selected"] > 0], want["normvec"][want["selected"] > 0])
        n, na, nn = m.map_incremental(fe, 0, st)
        ref2, wa, wn = oracle.map_incremental(ref, down, st, st)
        assert (n, na, nn) == (len(ref2), wa, wn)
        assert np.array_equal(canon(m.points()), canon(ref2))
        ref = ref2
        tree = oracle.KdTree(ref)


def test_map_incremental_refuses_stale_neighbour_indices(pkg, oracle, synthetic):
    """map_incremental replays neighbour INDICES of the scan slot's feature extraction: the entry points refuse when the map is another one or has
    been renumbered since (a second map_incremental, Delete_Point_Boxes, Build), and a scan slot twice in one batch."""
    import torch
    S = 2
    fe = pkg.LidarFrontEnd(max_points_per_scan=140000, max_scans=S)
    scene = synthetic.Scene(3)
    raws = [synthetic.lidar_scan(scene, f) for f in (1, 2)]
    states = np.stack([pkg.pack_lidar_state(*synthetic.lidar_state(f)[:2]) for f in (1, 2)])
    street = synthetic.lidar_map(scene, x_from=-40.0, x_to=60.0)
    maps = [pkg.LidarMap(), pkg.LidarMap()]
    for m in maps:
        m.Build(street)
    raw = np.concatenate(raws)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in raws])]).astype(np.int32)
    dev = torch.from_numpy(raw.view(np.uint8)).cuda()
    fe.frontend_batch(dev.data_ptr(), offs, maps, states, want_points=False)
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.map_incremental_batch(fe, [0, 0], maps, states)          # a scan slot twice
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.map_incremental_batch(fe, [0, 1], maps[::-1], states)    # slot 0 was matched against maps[0], not maps[1]
    size0 = maps[0].size()
    na, nn, sz = pkg.capi.map_incremental_batch(fe, [0, 1], maps, states)
    assert na[0] + nn[0] > 0 and sz[0] > size0
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.map_incremental_batch(fe, [0, 1], maps, states)          # the maps have been renumbered by the call above
    fe.frontend_batch(dev.data_ptr(), offs, maps, states, want_points=False)
    pkg.capi.delete_point_boxes_batch([maps[1]], [np.array([[-5, -5, -5, 5, 5, 5]], np.float32)])
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.map_incremental_batch(fe, [1], [maps[1]], states[1:])    # a box deletion in between
    na, _, _ = pkg.capi.map_incremental_batch(fe, [0], [maps[0]], states[:1])  # slot 0's map is untouched: fine
    assert na[0] >= 0
