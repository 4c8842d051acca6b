"""GPU parity of the HIP stereo matcher (Frame::ComputeStereoMatches) with the oracle: uRight, depth and SAD are
compared bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(pkg, oracle, synthetic, seed, w, h, nfeat=2000):
    left, right = synthetic.stereo_pair(seed, w, h)
    el = pkg.OrbExtractor(nfeatures=nfeat, max_width=w, max_height=h, max_images=1)
    er = pkg.OrbExtractor(nfeatures=nfeat, max_width=w, max_height=h, max_images=1)
    ol, orr = oracle.OrbOracle(nfeatures=nfeat), oracle.OrbOracle(nfeatures=nfeat)
    _, kl, dl = el.extract(left)
    _, kr, dr = er.extract(right)
    _, okl, odl = ol.extract(left)
    _, okr, odr = orr.extract(right)
    assert np.array_equal(dl, odl) and np.array_equal(dr, odr)
    return el, er, ol, orr, kl, dl, kr, dr


@pytest.mark.parametrize("seed,w,h", [(0, 1242, 375), (3, 1226, 370), (5, 640, 300)])
def test_stereo_single_frame(pkg, oracle, synthetic, seed, w, h):
    el, er, ol, orr, kl, dl, kr, dr = _pair(pkg, oracle, synthetic, seed, w, h)
    bf = np.float32(synthetic.BF)
    b = np.float32(bf / np.float32(synthetic.FX))
    want = oracle.stereo_match(ol, orr, kl, dl, kr, dr, float(bf), float(b))
    got = pkg.compute_stereo_matches(el, er, kl, dl, kr, dr, float(bf), float(b))
    assert np.array_equal(got[2], want[2]), "SAD"
    assert np.array_equal(got[0], want[0]), "uRight"
    assert np.array_equal(got[1], want[1]), "depth"
    assert (got[1] > 0).sum() > 0.3 * len(kl)


def test_stereo_edge_cases(pkg, oracle, synthetic):
    el, er, ol, orr, kl, dl, kr, dr = _pair(pkg, oracle, synthetic, 7, 800, 300, nfeat=800)
    bf, b = float(np.float32(synthetic.BF)), float(np.float32(synthetic.BF) / np.float32(synthetic.FX))
    # no right keypoints: nothing matches
    u, d, s = pkg.compute_stereo_matches(el, er, kl, dl, kr[:0], dr[:0], bf, b)
    assert np.all(u == -1) and np.all(d == -1) and np.all(s == -1)
    # no left keypoints
    u, d, s = pkg.compute_stereo_matches(el, er, kl[:0], dl[:0], kr, dr, bf, b)
    assert len(u) == 0
    # identical images left/right: every SAD is 0, so the median cut (threshold 0) rejects every match (Frame.cc:997-1010)
    left, _ = synthetic.stereo_pair(7, 800, 300)
    _, k2, d2 = er.extract(left)
    orr.extract(left)
    want = oracle.stereo_match(ol, orr, kl, dl, k2, d2, bf, b)
    got = pkg.compute_stereo_matches(el, er, kl, dl, k2, d2, bf, b)
    for g, w_ in zip(got, want):
        assert np.array_equal(g, w_)
    assert np.all(got[1] == -1) and (got[2] == 0).sum() > 100
    # a tiny baseline (small maxD) and a permuted right list (ties resolve by right index in both)
    perm = np.random.default_rng(0).permutation(len(kr))
    orr.extract(synthetic.stereo_pair(7, 800, 300)[1])
    er.extract(synthetic.stereo_pair(7, 800, 300)[1])
    want = oracle.stereo_match(ol, orr, kl, dl, kr[perm], dr[perm], bf, b * 8)
    got = pkg.compute_stereo_matches(el, er, kl, dl, kr[perm], dr[perm], bf, b * 8)
    for g, w_ in zip(got, want):
        assert np.array_equal(g, w_)


def test_stereo_batch(pkg, oracle, synthetic):
    import torch
    frames = synthetic.stereo_batch(3, seed=40)
    n = 6
    h, w = frames.shape[2:]
    dev = torch.from_numpy(frames.reshape(n, h, w)).cuda()
    e = pkg.OrbExtractor(max_width=w, max_height=h, max_images=n)
    kps, desc, counts, mono = e.extract_batch_dev(dev.data_ptr(), n, w, h, w, w * h)
    bf = np.float32(synthetic.BF)
    b = np.float32(bf / np.float32(synthetic.FX))
    u, d, s = pkg.stereo_match_batch(e, 3, float(bf), float(b))
    for f in range(3):
        ol, orr = oracle.OrbOracle(), oracle.OrbOracle()
        _, kl, dl = ol.extract(frames[f, 0])
        _, kr, dr = orr.extract(frames[f, 1])
        want = oracle.stereo_match(ol, orr, kl, dl, kr, dr, float(bf), float(b))
        nl = counts[2 * f]
        assert nl == len(kl)
        assert np.array_equal(s[f, :nl], want[2])
        assert np.array_equal(u[f, :nl], want[0])
        assert np.array_equal(d[f, :nl], want[1])
    e.close()
