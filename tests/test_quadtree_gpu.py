"""The device keypoint distribution (k_quadtree, csrc/quadtree_kernels.hip) against the oracle's restatement of
ORBextractor::DistributeOctTree (SF/src/ORBextractor.cc:529-753): same keypoints in the same order (bit-exact), for every workgroup
size the kernel is built for."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _candidates(seed, n, w, h, r_lo=7, r_hi=120):
    rng = np.random.default_rng(seed)
    pos = rng.choice(w * h, size=n, replace=False) if n else np.zeros(0, np.int64)
    pos.sort()
    return np.stack([pos % w, pos // w, rng.integers(r_lo, r_hi, n)], 1).astype(np.float32).reshape(-1, 3)


@pytest.mark.parametrize("threads", [0, -1, -2, 256, 512, 1024])  # 0 / -1 - c: the sorted-path form in LDS (default / from class c on), else the global-memory form
@pytest.mark.parametrize("seed,n,w,h,target", [(0, 5000, 1210, 343, 434), (1, 300, 315, 73, 122), (2, 40, 500, 200, 100), (3, 1, 400, 300, 50),
                                               (4, 0, 400, 300, 50), (5, 2500, 640, 640, 700), (6, 900, 980, 260, 1), (7, 20000, 1210, 343, 434),
                                               (8, 3, 1210, 343, 434), (9, 2000, 900, 100, 300), (10, 700, 1210, 343, 2000),
                                               # the LDS classes of the sorted form: ~3 k / ~6 k / ~12 k candidates, then the global-memory form
                                               (11, 2900, 1210, 343, 434), (12, 6100, 1210, 343, 434), (13, 11500, 1210, 343, 434),
                                               (14, 13500, 1210, 343, 434), (15, 4000, 2000, 150, 600)])
def test_device_quadtree_equals_oracle(pkg, oracle, threads, seed, n, w, h, target):
    xyr = _candidates(seed, n, w, h)
    want = oracle.OrbOracle().distribute(xyr, 16, 16 + w, 16, 16 + h, target)
    got = pkg.distribute_quadtree_device(xyr, 16, 16 + w, 16, 16 + h, target, threads)
    assert np.array_equal(got, want)
    assert np.array_equal(got, pkg.distribute_quadtree_host(xyr, 16, 16 + w, 16, 16 + h, target))


@pytest.mark.parametrize("seed", range(6))
def test_device_quadtree_ties_in_the_closing_sort(pkg, oracle, seed):
    """Clustered candidates on a coarse lattice: many nodes with equal population and equal UL.x meet in the closing phase, where the
    reference's std::sort (not stable) decides the division order -- the kernel restates libstdc++'s introsort move for move."""
    rng = np.random.default_rng(100 + seed)
    w, h = 1210, 343
    gx, gy = np.meshgrid(np.arange(4, w, 9), np.arange(4, h, 9))
    base = np.stack([gx.ravel(), gy.ravel()], 1)
    pts = np.concatenate([base, base + [1, 0], base[rng.random(len(base)) < 0.5] + [0, 2]])
    pts = np.unique(pts, axis=0)
    pts = pts[np.lexsort((pts[:, 0], pts[:, 1]))]
    xyr = np.concatenate([pts, rng.integers(7, 12, (len(pts), 1))], 1).astype(np.float32)
    for target in (200, 434, 1500, 3000):
        want = oracle.OrbOracle().distribute(xyr, 16, 16 + w, 16, 16 + h, target)
        got = pkg.distribute_quadtree_device(xyr, 16, 16 + w, 16, 16 + h, target)
        assert np.array_equal(got, want), target


def test_device_quadtree_on_real_candidates(pkg, oracle, synthetic):
    left, _ = synthetic.stereo_pair(7, 800, 300)
    o = oracle.OrbOracle()
    o.extract(left)
    per_level = o.tables()[1]
    for lvl in range(8):
        c = o.candidates(lvl).copy()
        lh, lw = o.level(lvl).shape
        c[:, :2] -= 16
        want = o.distribute(c, 16, lw - 16, 16, lh - 16, int(per_level[lvl]))
        got = pkg.distribute_quadtree_device(c, 16, lw - 16, 16, lh - 16, int(per_level[lvl]))
        assert np.array_equal(got, want), lvl
