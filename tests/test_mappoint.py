"""Map-point refresh (MapPoint::ComputeDistinctiveDescriptors + UpdateNormalAndDepth): properties of
the oracle on the CPU, bit-exact parity of the product on the GPU."""
import numpy as np
import pytest


def problem(seed, n_points=300, max_obs=40):
    rng = np.random.default_rng(seed)
    counts = rng.integers(0, max_obs + 1, n_points)
    counts[:5] = [0, 1, 2, 3, max_obs]
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    total = int(off[-1])
    desc = np.zeros((total, 32), np.uint8)
    for p in range(n_points):
        base = rng.integers(0, 256, 32).astype(np.uint8)
        for k in range(off[p], off[p + 1]):
            d = base.copy()
            for bit in rng.choice(256, size=int(rng.integers(0, 60)), replace=False):
                d[bit // 8] ^= np.uint8(1 << (bit % 8))
            desc[k] = d
    pos = rng.normal(0, 10, (n_points, 3)).astype(np.float32)
    centres = (np.repeat(pos, counts, 0) + rng.normal(0, 8, (total, 3))).astype(np.float32)
    ref = (pos + rng.normal(0, 8, (n_points, 3))).astype(np.float32)
    scales = (np.float32(1.2) ** rng.integers(0, 8, n_points)).astype(np.float32)
    return off, desc, centres, pos, ref, scales, np.float32(1.2) ** 7


def test_oracle_properties(oracle):
    off, desc, centres, pos, ref, scales, last = problem(0)
    best, normals, mn, mx = oracle.map_points_refresh(off, desc, centres, pos, ref, scales, last)
    assert best[0] == -1 and best[1] == 0 and best[2] == 0  # no observations; one; two (both medians equal: the first wins)
    for p in range(3, 40):
        b, n = off[p], off[p + 1] - off[p]
        if n == 0:
            continue
        D = np.array([[np.unpackbits(desc[b + i] ^ desc[b + j]).sum() for j in range(n)] for i in range(n)])
        med = np.sort(D, 1)[:, int(0.5 * (n - 1))]
        assert best[p] == int(np.argmin(med))
        v = pos[p] - centres[b:b + n]
        want = (v / np.linalg.norm(v, axis=1, keepdims=True)).mean(0)
        assert np.allclose(normals[p], want, atol=1e-5)
        d = np.linalg.norm(pos[p] - ref[p])
        assert np.isclose(mx[p], d * scales[p], rtol=1e-6) and np.isclose(mn[p], mx[p] / last, rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_points,max_obs", [(1, 400, 40), (2, 50, 112), (3, 2000, 12)])
def test_product_matches_the_oracle(pkg, oracle, seed, n_points, max_obs):
    off, desc, centres, pos, ref, scales, last = problem(seed, n_points, max_obs)
    want = oracle.map_points_refresh(off, desc, centres, pos, ref, scales, last)
    got = pkg.capi.map_points_refresh(off, desc, centres, pos, ref, scales, last)
    assert np.array_equal(got[0], want[0])
    has = want[0] >= 0
    for g, w in zip(got[1:], want[1:]):
        assert np.array_equal(g[has], w[has])  # float arithmetic in the reference's order: bit for bit
    assert np.all(got[1][~has] == 0)


@pytest.mark.gpu
def test_product_edge_cases(pkg, oracle):
    off, desc, centres, pos, ref, scales, last = problem(4, 20, 10)
    with pytest.raises(pkg.capi.Tc2liError):  # more observations than the kernel's table holds
        big = np.array([0, 113], np.int32)
        pkg.capi.map_points_refresh(big, np.zeros((113, 32), np.uint8), np.zeros((113, 3), np.float32), pos[:1], ref[:1], scales[:1], last)
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.map_points_refresh(off, desc, centres, pos, ref, scales, 0.0)
    empty = pkg.capi.map_points_refresh(np.zeros(1, np.int32), np.zeros((0, 32), np.uint8), np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32),
                                        np.zeros((0, 3), np.float32), np.zeros(0, np.float32), last)
    assert len(empty[0]) == 0
    # identical descriptors: every median is 0, the first observation wins
    same = np.tile(desc[:1], (7, 1))
    best = pkg.capi.map_points_refresh(np.array([0, 7], np.int32), same, centres[:7], pos[:1], ref[:1], scales[:1], last)[0]
    assert best[0] == 0
