"""GPU parity of the projection matcher with the oracle: identical query construction (host float arithmetic) and
identical match assignments for both tracking overloads."""
import numpy as np
import pytest

from matcher_scenario import make, local_map_points

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sc(oracle, synthetic):
    return make(oracle, synthetic, seed=1)


def to_pkg_keys(pkg, k):
    out = np.zeros(len(k), pkg.capi.KEYPOINT_DTYPE) if hasattr(pkg, "capi") else None
    return k


@pytest.mark.parametrize("th,mono", [(7.0, False), (14.0, False), (7.0, True)])
def test_last_frame_overload(pkg, oracle, sc, th, mono):
    args = (sc["pose_cur"], sc["pose_last"], sc["cam4"], sc["b"], sc["bf"], sc["scales"], sc["cols"], sc["rows"], sc["has_point"],
            sc["outlier"], sc["Xw"], sc["last_keys"], sc["mp_desc"], th, mono)
    want_q = oracle.project_last_frame(*args)
    got_q = pkg.project_last_frame(*args)
    assert got_q.tobytes() == want_q.tobytes()
    for check in (False, True):
        wn, wm = oracle.search_by_projection(sc["keys"], sc["desc"], sc["u_right"], sc["occupied"], sc["cols"], sc["rows"], want_q, 0, 0.9, check)
        gn, gm, gk = pkg.search_by_projection(sc["keys"], sc["desc"], sc["u_right"], sc["occupied"], sc["cols"], sc["rows"], got_q, 0, 0.9, check)
        assert gn == wn and np.array_equal(gm, wm) and gn > 300
        assert np.array_equal(np.sort(gk[gk >= 0]), np.nonzero(gm >= 0)[0])


@pytest.mark.parametrize("th", [1.0, 3.0, 10.0])
def test_local_map_overload(pkg, oracle, sc, th):
    rng = np.random.default_rng(11)
    pts = local_map_points(sc, oracle, rng)
    log_s = float(np.log(np.float32(1.2)))
    want_q = oracle.project_local_map(sc["pose_cur"], sc["cam4"], sc["bf"], sc["scales"], log_s, sc["cols"], sc["rows"], pts, th, True, 60.0)
    got_q = pkg.project_local_map(sc["pose_cur"], sc["cam4"], sc["bf"], sc["scales"], log_s, sc["cols"], sc["rows"], pts, th, True, 60.0)
    assert got_q.tobytes() == want_q.tobytes()
    wn, wm = oracle.search_by_projection(sc["keys"], sc["desc"], sc["u_right"], sc["occupied"], sc["cols"], sc["rows"], want_q, 1, 0.8)
    gn, gm, _ = pkg.search_by_projection(sc["keys"], sc["desc"], sc["u_right"], sc["occupied"], sc["cols"], sc["rows"], got_q, 1, 0.8)
    assert gn == wn and np.array_equal(gm, wm) and gn > 300


def test_matcher_edge_cases(pkg, oracle, sc):
    q = oracle.project_last_frame(sc["pose_cur"], sc["pose_last"], sc["cam4"], sc["b"], sc["bf"], sc["scales"], sc["cols"], sc["rows"],
                                  sc["has_point"], sc["outlier"], sc["Xw"], sc["last_keys"], sc["mp_desc"], 7.0)
    # heavy contention: every query carries the same descriptor and a huge window
    qc = q.copy()
    qc["descriptor"][:] = qc["descriptor"][0]
    qc["radius"] = 60.0; qc["min_level"] = -1; qc["max_level"] = -1
    wn, wm = oracle.search_by_projection(sc["keys"], sc["desc"], sc["u_right"], None, sc["cols"], sc["rows"], qc, 0, 0.9)
    gn, gm, _ = pkg.search_by_projection(sc["keys"], sc["desc"], sc["u_right"], None, sc["cols"], sc["rows"], qc, 0, 0.9)
    assert gn == wn and np.array_equal(gm, wm)
    # queries whose map points have no observations do not block later ones
    qn = q.copy(); qn["has_observations"][::2] = 0
    wn, wm = oracle.search_by_projection(sc["keys"], sc["desc"], sc["u_right"], None, sc["cols"], sc["rows"], qn, 0, 0.9)
    gn, gm, _ = pkg.search_by_projection(sc["keys"], sc["desc"], sc["u_right"], None, sc["cols"], sc["rows"], qn, 0, 0.9)
    assert gn == wn and np.array_equal(gm, wm)
    # no queries / no keypoints / everything occupied
    assert pkg.search_by_projection(sc["keys"], sc["desc"], sc["u_right"], None, sc["cols"], sc["rows"], q[:0], 0)[0] == 0
    assert pkg.search_by_projection(sc["keys"][:0], sc["desc"][:0], sc["u_right"][:0], None, sc["cols"], sc["rows"], q, 0)[0] == 0
    full = np.ones(len(sc["keys"]), np.uint8)
    assert pkg.search_by_projection(sc["keys"], sc["desc"], sc["u_right"], full, sc["cols"], sc["rows"], q, 0)[0] == 0
