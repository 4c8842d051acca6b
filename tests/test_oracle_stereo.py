"""CPU tests of the stereo/grid oracle: Hamming distance against numpy, the stereo depths against the depth map
the synthetic renderer knows exactly, GetFeaturesInArea against a brute-force statement."""
import numpy as np


def test_descriptor_distance(oracle):
    rng = np.random.default_rng(0)
    for _ in range(200):
        a, b = rng.integers(0, 256, (2, 32)).astype(np.uint8)
        assert oracle.descriptor_distance(a, b) == int(np.unpackbits(a ^ b).sum())
    assert oracle.descriptor_distance(a, a) == 0


def test_stereo_depth_against_rendered_truth(oracle, synthetic):
    sc = synthetic.Scene(4)
    left, depth = sc.render(0.0, noise_seed=1)
    right, _ = sc.render(synthetic.BASELINE, noise_seed=2)
    ol, orr = oracle.OrbOracle(), oracle.OrbOracle()
    _, kl, dl = ol.extract(left)
    _, kr, dr = orr.extract(right)
    mbf = np.float32(synthetic.BF)
    mb = np.float32(mbf / np.float32(synthetic.FX))
    u, d, s = oracle.stereo_match(ol, orr, kl, dl, kr, dr, float(mbf), float(mb))
    ok = d > 0
    assert ok.sum() > 600
    assert np.all((u >= 0) == ok)
    # uRight = uL - bf/depth by construction (Frame.cc:985-993)
    disp = kl["x"][ok] - u[ok]
    assert np.all(disp > 0) and np.allclose(np.float32(mbf) / disp, d[ok], rtol=1e-6)
    truth = depth[np.rint(kl["y"][ok]).astype(int), np.rint(kl["x"][ok]).astype(int)]
    rel = np.abs(d[ok] - truth) / truth
    assert np.median(rel) < 0.02          # sub-pixel disparity on exact synthetic geometry
    assert (rel < 0.1).mean() > 0.85      # a few keypoints sit on depth discontinuities
    # the median cut removed the worst SADs: survivors are all below 1.5*1.4*median
    acc = s[s >= 0]
    med = np.sort(acc)[len(acc) // 2]
    assert np.all(s[ok] < np.float32(1.5) * np.float32(1.4) * np.float32(med))
    assert np.all(~ok[(s >= 0) & (s >= np.float32(1.5) * np.float32(1.4) * np.float32(med))])


def test_features_in_area(oracle, synthetic):
    left, _ = synthetic.stereo_pair(9, 800, 300)
    o = oracle.OrbOracle(nfeatures=1000)
    _, k, _ = o.extract(left)
    rng = np.random.default_rng(1)
    for _ in range(50):
        x, y = rng.uniform(0, 800), rng.uniform(0, 300)
        r = rng.uniform(5, 60)
        lo, hi = (-1, -1) if rng.random() < 0.5 else (int(rng.integers(0, 4)), int(rng.integers(3, 8)))
        got = oracle.features_in_area(k, 800, 300, x, y, r, lo, hi)
        m = (np.abs(k["x"] - np.float32(x)) < np.float32(r)) & (np.abs(k["y"] - np.float32(y)) < np.float32(r))
        if lo > 0 or hi >= 0:
            m &= k["octave"] >= lo
            if hi >= 0:
                m &= k["octave"] <= hi
        # the grid may miss nothing: PosInGrid rounds to the nearest cell while the query uses floor/ceil bounds
        assert set(got.tolist()) <= set(np.nonzero(m)[0].tolist())
        assert len(got) >= 0.9 * m.sum() - 2
