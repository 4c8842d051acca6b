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
