"""CPU tests of the projection-matching oracle: query construction against float64 numpy geometry, the greedy
matcher against an independent pure-python statement of the loops, the rotation filter."""
import numpy as np
import pytest

from matcher_scenario import make, local_map_points


@pytest.fixture(scope="module")
def sc(oracle, synthetic):
    return make(oracle, synthetic, seed=3, w=800, h=300, nfeat=1000)


def quat_R(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def test_last_frame_queries(oracle, sc):
    q = oracle.project_last_frame(sc["pose_cur"], sc["pose_last"], sc["cam4"], sc["b"], sc["bf"], sc["scales"], sc["cols"], sc["rows"],
                                  sc["has_point"], sc["outlier"], sc["Xw"], sc["last_keys"], sc["mp_desc"], 7.0)
    R, t = quat_R(sc["pose_cur"][:4]), sc["pose_cur"][4:].astype(np.float64)
    pc = sc["Xw"].astype(np.float64) @ R.T + t
    u = sc["cam4"][0] * pc[:, 0] / pc[:, 2] + sc["cam4"][2]
    v = sc["cam4"][1] * pc[:, 1] / pc[:, 2] + sc["cam4"][3]
    ok = (sc["has_point"] > 0) & (sc["outlier"] == 0) & (pc[:, 2] > 0) & (u >= 0) & (u <= sc["cols"]) & (v >= 0) & (v <= sc["rows"])
    border = (np.abs(u) < 1e-3) | (np.abs(u - sc["cols"]) < 1e-3) | (np.abs(v) < 1e-3) | (np.abs(v - sc["rows"]) < 1e-3)
    assert np.array_equal(q["valid"][~border] > 0, ok[~border])
    m = q["valid"] > 0
    assert np.allclose(q["u"][m], u[m], atol=2e-3) and np.allclose(q["v"][m], v[m], atol=2e-3)
    assert np.allclose(q["u_right"][m], u[m] - sc["bf"] / pc[m, 2], atol=5e-3)
    assert np.allclose(q["radius"][m], 7.0 * sc["scales"][sc["last_keys"]["octave"][m]])
    # the camera moved forward by less than the baseline: neither forward nor backward -> levels oct-1 .. oct+1
    assert np.array_equal(q["min_level"][m], sc["last_keys"]["octave"][m] - 1)
    assert np.array_equal(q["max_level"][m], sc["last_keys"]["octave"][m] + 1)
    fwd = sc["pose_cur"].copy(); fwd[6] = -2.0  # tlc.z = +2 > mb: forward
    q2 = oracle.project_last_frame(fwd, sc["pose_last"], sc["cam4"], sc["b"], sc["bf"], sc["scales"], sc["cols"], sc["rows"],
                                   sc["has_point"], sc["outlier"], sc["Xw"], sc["last_keys"], sc["mp_desc"], 7.0)
    m2 = q2["valid"] > 0
    assert m2.sum() > 10 and np.all(q2["max_level"][m2] == -1) and np.array_equal(q2["min_level"][m2], sc["last_keys"]["octave"][m2])


def reference_greedy(sc, queries, mode, ratio, occupied):
    """Independent statement of the matching loops (pure python, grid-free: the grid only bounds the candidate set)."""
    keys, desc, ur = sc["keys"], sc["desc"], sc["u_right"]
    taken = occupied.astype(bool).copy()
    gw, gh = np.float32(64) / np.float32(sc["cols"]), np.float32(48) / np.float32(sc["rows"])
    px = np.round(keys["x"] * gw).astype(int); py = np.round(keys["y"] * gh).astype(int)
    ingrid = (px >= 0) & (px < 64) & (py >= 0) & (py < 48)
    out = np.full(len(queries), -1, int)
    bits = np.unpackbits(desc, axis=1)
    for qi, Q in enumerate(queries):
        if not Q["valid"]:
            continue
        r = Q["radius"]
        c0x = max(0, int(np.floor((Q["u"] - r) * gw))); c1x = min(63, int(np.ceil((Q["u"] + r) * gw)))
        c0y = max(0, int(np.floor((Q["v"] - r) * gh))); c1y = min(47, int(np.ceil((Q["v"] + r) * gh)))
        cand = np.nonzero(ingrid & (px >= c0x) & (px <= c1x) & (py >= c0y) & (py <= c1y) & (np.abs(keys["x"] - Q["u"]) < r) &
                          (np.abs(keys["y"] - Q["v"]) < r))[0]
        if Q["min_level"] > 0 or Q["max_level"] >= 0:
            cand = cand[keys["octave"][cand] >= Q["min_level"]]
            if Q["max_level"] >= 0:
                cand = cand[keys["octave"][cand] <= Q["max_level"]]
        cand = sorted(cand, key=lambda i: (px[i], py[i], i))  # ix outer, iy inner, insertion order inside a cell
        qb = np.unpackbits(Q["descriptor"])
        best, best2, bl, bl2, bi = 256, 256, -1, -1, -1
        for i in cand:
            if taken[i]:
                continue
            if ur[i] > 0 and abs(np.float32(Q["u_right"]) - ur[i]) > r:
                continue
            d = int((bits[i] != qb).sum())
            if mode == 0:
                if d < best:
                    best, bi = d, i
            elif d < best:
                best2, best, bl2, bl, bi = best, d, bl, keys["octave"][i], i
            elif d < best2:
                bl2, best2 = keys["octave"][i], d
        if best <= 100:
            if mode == 1 and bl == bl2 and np.float32(best) > np.float32(ratio) * np.float32(best2):
                continue
            out[qi] = bi
            if Q["has_observations"]:
                taken[bi] = True
    return out


@pytest.mark.parametrize("mode,th", [(0, 7.0), (0, 15.0), (1, 1.0), (1, 3.0)])
def test_greedy_matching(oracle, sc, mode, th):
    rng = np.random.default_rng(7)
    if mode == 0:
        q = oracle.project_last_frame(sc["pose_cur"], sc["pose_last"], sc["cam4"], sc["b"], sc["bf"], sc["scales"], sc["cols"], sc["rows"],
                                      sc["has_point"], sc["outlier"], sc["Xw"], sc["last_keys"], sc["mp_desc"], th)
    else:
        pts = local_map_points(sc, oracle, rng)
        q = oracle.project_local_map(sc["pose_cur"], sc["cam4"], sc["bf"], sc["scales"], float(np.log(np.float32(1.2))), sc["cols"], sc["rows"],
                                     pts, th)
    n, match = oracle.search_by_projection(sc["keys"], sc["desc"], sc["u_right"], sc["occupied"], sc["cols"], sc["rows"], q, mode, 0.8)
    want = reference_greedy(sc, q, mode, 0.8, sc["occupied"])
    assert np.array_equal(match, want)
    assert n == (want >= 0).sum() and n > 100
    m = match[match >= 0]
    assert len(np.unique(m)) == len(m)                # a keypoint is matched at most once
    assert not np.any(sc["occupied"][m])              # and never one that was occupied before
    if mode == 0:                                     # most matches are the true correspondences
        src = np.nonzero(match >= 0)[0]
        assert (sc["order"][src] == match[src]).mean() > 0.9


def test_rotation_filter(oracle, sc):
    q = oracle.project_last_frame(sc["pose_cur"], sc["pose_last"], sc["cam4"], sc["b"], sc["bf"], sc["scales"], sc["cols"], sc["rows"],
                                  sc["has_point"], sc["outlier"], sc["Xw"], sc["last_keys"], sc["mp_desc"], 7.0)
    q["angle"][::7] = (q["angle"][::7] + np.float32(100)) % np.float32(360)  # inconsistent rotations
    n0, m0 = oracle.search_by_projection(sc["keys"], sc["desc"], sc["u_right"], None, sc["cols"], sc["rows"], q, 0, 0.9, False)
    n1, m1 = oracle.search_by_projection(sc["keys"], sc["desc"], sc["u_right"], None, sc["cols"], sc["rows"], q, 0, 0.9, True)
    assert n1 < n0 and n1 == (m1 >= 0).sum()
    rot = (q["angle"] - sc["keys"]["angle"][np.maximum(m0, 0)]) % 360
    bins = np.round(rot / 30).astype(int) % 12  # coarse check: survivors sit in few bins
    assert len(np.unique(bins[m1 >= 0])) <= 4
    assert np.all(m1[m0 < 0] == -1)
