"""Local-map bookkeeping: Tracking::UpdateLocalKeyFrames + UpdateLocalPoints on flat arrays.  The oracle restates
the reference's loops (including its early exits); the product's kernels must give the same lists in the same order."""
import numpy as np
import pytest


def random_graph(seed, K, P, slots=300, bad_frac=0.05, max_obs=9, chain=False):
    rng = np.random.default_rng(seed)
    kf_bad = (rng.random(K) < bad_frac).astype(np.uint8)
    point_bad = (rng.random(P) < bad_frac).astype(np.uint8)

    def csr(lists):
        off = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.int32)
        val = np.concatenate([np.asarray(l, np.int32) for l in lists]) if off[-1] else np.zeros(0, np.int32)
        return off, val.astype(np.int32)

    # keyframes see points near "their" stretch of the map, so neighbours share points as in a trajectory
    centre = np.linspace(0, P, K)
    matches = []
    for k in range(K):
        ids = np.clip((centre[k] + rng.normal(0, max(P / K * 4, 8), slots)).astype(int), 0, P - 1)
        ids[rng.random(slots) < 0.4] = -1
        matches.append(ids)
    obs = [[] for _ in range(P)]
    for k in range(K):
        for p in np.unique(matches[k][matches[k] >= 0]):
            if len(obs[p]) < max_obs:
                obs[p].append(k)
    covis = []
    for k in range(K):
        n = rng.integers(0, 16)
        cand = np.clip(k + rng.permutation(np.arange(-12, 13))[:n], 0, K - 1)
        covis.append([c for c in dict.fromkeys(cand.tolist()) if c != k])
    parent = np.full(K, -1, np.int32)
    for k in range(1, K):
        parent[k] = k - 1 if chain else rng.integers(max(0, k - 6), k)
    children = [sorted(np.nonzero(parent == k)[0].tolist()) for k in range(K)]
    prev_kf = np.arange(-1, K - 1).astype(np.int32)
    co, cv = csr(covis); ho, hv = csr(children); mo, mv = csr(matches); oo, ov = csr(obs)
    return dict(kf_bad=kf_bad, covis_off=co, covis=cv, child_off=ho, children=hv, parent=parent, prev_kf=prev_kf, match_off=mo, matches=mv,
                point_bad=point_bad, obs_off=oo, obs_kf=ov)


def frame_points_near(rng, g, kf, n=400):
    m = g["matches"][g["match_off"][kf]:g["match_off"][kf + 1]]
    fp = rng.choice(m, n)
    fp[rng.random(n) < 0.3] = -1
    return fp.astype(np.int32)


def test_oracle_local_map_by_hand(oracle):
    """Three keyframes in a chain; the frame sees points of keyframe 1 -> votes, the parent extension and its early exit."""
    g = dict(kf_bad=[0, 0, 0, 0], covis_off=[0, 1, 3, 4, 4], covis=[1, 0, 2, 1], child_off=[0, 1, 2, 3, 3], children=[1, 2, 3], parent=[-1, 0, 1, 2],
             prev_kf=[-1, 0, 1, 2], match_off=[0, 2, 5, 7, 8], matches=[0, 1, 1, 2, -1, 2, 3, 4], point_bad=[0, 0, 0, 1, 0], obs_off=[0, 1, 3, 5, 6, 7],
             obs_kf=[0, 0, 1, 1, 2, 2, 3])
    kfs, ref, pts, cleared = oracle.update_local_map(g, [1, 2, 3, -1])
    # point 1 votes {0, 1}, point 2 votes {1, 2}, point 3 is bad (cleared): counter {0: 1, 1: 2, 2: 1}; keyframe 0 adds nothing through
    # covisibility (1 is marked) or children (1 marked); keyframe 1: child 2 marked, parent 0 marked; keyframe 2: child 3 is added, then its
    # parent 1 is marked -> no break; reference = 1
    assert kfs.tolist() == [0, 1, 2, 3] and ref == 1 and cleared.tolist() == [False, False, True, False]
    # backwards: kf 3 -> point 4; kf 2 -> 2 (3 is bad); kf 1 -> 1 (2 seen); kf 0 -> 0
    assert pts.tolist() == [4, 2, 1, 0]
    # the temporal block only runs below 80 keyframes and stops at the first marked keyframe
    kfs2, _, _, _ = oracle.update_local_map(g, [0], temporal_last_kf=3)
    assert kfs2.tolist() == [0, 1, 3, 2]  # vote {0}; covisibility adds 1; child 1 marked; no parent; temporal: 3, 2, then 1 is marked


CASES = [(0, 40, 900, -1), (1, 150, 4000, -1), (2, 150, 4000, 149), (3, 600, 20000, 300), (4, 12, 100, 5), (5, 2300, 30000, -1)]


@pytest.mark.gpu
@pytest.mark.parametrize("seed,K,P,temporal", CASES)
def test_local_map_update_equals_oracle(pkg, oracle, seed, K, P, temporal):
    g = random_graph(seed, K, P, chain=seed % 2 == 0)
    lm = pkg.capi.LocalMap()
    lm.set_graph(g)
    rng = np.random.default_rng(100 + seed)
    sizes = []
    for kf in [0, K // 3, K // 2, K - 1]:
        fp = frame_points_near(rng, g, kf)
        want = oracle.update_local_map(g, fp, temporal)
        got = lm.update(fp, temporal)
        assert np.array_equal(got[0], want[0]), "local keyframes"
        assert got[1] == want[1]
        assert np.array_equal(got[2], want[2]), "local points"
        assert np.array_equal(got[3], want[3])
        sizes.append((len(want[0]), len(want[2])))
    assert max(s[0] for s in sizes) > 5 and max(s[1] for s in sizes) > 50
    lm.close()


@pytest.mark.gpu
def test_local_map_edge_cases(pkg, oracle):
    g = random_graph(9, 30, 500)
    lm = pkg.capi.LocalMap()
    with pytest.raises(pkg.capi.Tc2liError):
        lm.update(np.zeros(3, np.int32))          # no graph yet
    lm.set_graph(g)
    # a frame without any map point: no votes, no reference keyframe, empty lists (the temporal block still runs)
    for temporal in (-1, 7):
        fp = np.full(50, -1, np.int32)
        want = oracle.update_local_map(g, fp, temporal)
        got = lm.update(fp, temporal)
        assert got[1] == want[1] == -1 and np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])
    got = lm.update(np.zeros(0, np.int32))
    assert len(got[0]) == 0 and len(got[2]) == 0
    # only bad points: all cleared
    bad = np.nonzero(g["point_bad"])[0].astype(np.int32)
    got = lm.update(bad)
    assert got[3].all() and len(got[0]) == 0
    with pytest.raises(pkg.capi.Tc2liError):
        lm.update(np.array([500], np.int32))      # point index out of range
    fp = frame_points_near(np.random.default_rng(1), g, 10)
    with pytest.raises(pkg.capi.Tc2liError):
        lm.update(fp, keyframe_capacity=1)        # capacity
    with pytest.raises(pkg.capi.Tc2liError):
        lm.update(fp, point_capacity=3)
    # a new graph replaces the mirror
    g2 = random_graph(10, 45, 700)
    lm.set_graph(g2)
    fp = frame_points_near(np.random.default_rng(2), g2, 20)
    want = oracle.update_local_map(g2, fp)
    got = lm.update(fp)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])
    bad_graph = dict(g2); bad_graph["covis"] = g2["covis"].copy(); bad_graph["covis"][0] = 99
    with pytest.raises(pkg.capi.Tc2liError):
        lm.set_graph(bad_graph)
