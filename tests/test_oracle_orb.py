"""CPU tests of the oracle itself (no GPU): the restated OpenCV primitives against independent numpy statements
of their published definitions, and the oracle against the committed golden vectors."""
import math
import os

import numpy as np
import pytest

CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
          (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def brute_fast(img, th):
    """FAST-9/16 with score and strict 3x3 NMS straight from the definition (pure numpy/python, small images)."""
    h, w = img.shape
    im = img.astype(np.int32)
    score = np.zeros((h, w), np.int32)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            v = im[y, x]
            d = [v - im[y + dy, x + dx] for dx, dy in CIRCLE]
            best = -999
            for k in range(16):
                arc = [d[(k + m) % 16] for m in range(9)]
                best = max(best, min(arc), min(-a for a in arc))
            if best > th:
                score[y, x] = best - 1
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = score[y, x]
            if s == 0 and not (img[y, x] is None):
                # score 0 means "not a corner" unless th == 0 and best == 1; th >= 1 in every test
                continue
            nb = score[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if s > nb.max():
                out.append((x, y, s))
    return np.array(out, np.float32).reshape(-1, 3)


@pytest.mark.parametrize("seed,th", [(0, 20), (1, 7), (2, 12)])
def test_fast_against_definition(oracle, seed, th):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (41, 47)).astype(np.uint8)
    # add structure: smooth background with bright/dark squares
    img = (img // 8 + 100).astype(np.uint8)
    for _ in range(12):
        x, y = rng.integers(3, 40), rng.integers(3, 34)
        img[y:y + rng.integers(2, 7), x:x + rng.integers(2, 7)] = rng.integers(0, 256)
    got = oracle.fast9_16(img, th)
    want = brute_fast(img, th)
    assert len(want) > 0
    assert np.array_equal(got, want)


def test_fast_atan2_accuracy(oracle):
    rng = np.random.default_rng(0)
    for _ in range(2000):
        y, x = rng.integers(-200000, 200000, 2)
        a = oracle.fast_atan2(y, x)
        ref = math.degrees(math.atan2(y, x)) % 360.0
        assert abs((a - ref + 180) % 360 - 180) < 0.3  # cv::fastAtan2 documents ~0.3 degree accuracy
    assert oracle.fast_atan2(0, 0) == 0.0
    assert oracle.fast_atan2(0, 5) == 0.0


def test_resize_properties(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (60, 90)).astype(np.uint8)
    assert np.array_equal(oracle.resize_linear(img, 90, 60), img)  # identity size: weights (2048, 0)
    flat = np.full((50, 70), 137, np.uint8)
    assert np.all(oracle.resize_linear(flat, 58, 42) == 137)
    # independent float statement of bilinear sampling stays within 1 grey level of the fixed-point result
    small = oracle.resize_linear(img, 75, 50).astype(np.float64)
    sx = (np.arange(75) + 0.5) * (90 / 75) - 0.5
    sy = (np.arange(50) + 0.5) * (60 / 50) - 0.5
    x0 = np.clip(np.floor(sx).astype(int), 0, 89); x1 = np.clip(x0 + 1, 0, 89); fx = np.clip(sx - x0, 0, 1)
    y0 = np.clip(np.floor(sy).astype(int), 0, 59); y1 = np.clip(y0 + 1, 0, 59); fy = np.clip(sy - y0, 0, 1)
    f = img.astype(np.float64)
    ref = ((f[np.ix_(y0, x0)] * (1 - fx) + f[np.ix_(y0, x1)] * fx) * (1 - fy)[:, None] +
           (f[np.ix_(y1, x0)] * (1 - fx) + f[np.ix_(y1, x1)] * fx) * fy[:, None])
    assert np.abs(small - ref).max() <= 1.0


def test_gaussian_blur(oracle):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (33, 45)).astype(np.uint8)
    got = oracle.gaussian_blur7(img).astype(np.float64)
    k = np.array([18, 34, 48, 56, 48, 34, 18], np.float64) / 256
    pad = np.pad(img.astype(np.float64), 3, mode="reflect")  # numpy 'reflect' == BORDER_REFLECT_101
    hz = sum(k[i] * pad[:, i:i + 45] for i in range(7))
    ref = sum(k[i] * hz[i:i + 33, :] for i in range(7))
    assert np.array_equal(got, np.floor(ref + 0.5))
    assert np.all(oracle.gaussian_blur7(np.full((20, 20), 201, np.uint8)) == 201)
    # taps: error-diffused 8.8 quantisation of exp(-x^2/8), sum 256
    w = np.exp(-np.arange(-3, 4) ** 2 / 8.0); w /= w.sum()
    assert np.abs(w * 256 - np.array([18, 34, 48, 56, 48, 34, 18])).max() < 1.0


def test_ctor_tables(oracle):
    o = oracle.OrbOracle(2000, 1.2, 8, 20, 7)
    scale, per_level, umax = o.tables()
    assert list(per_level) == [434, 362, 302, 251, 209, 175, 145, 122]
    assert per_level.sum() == 2000
    assert list(umax) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert scale[0] == 1.0 and abs(scale[7] - 1.2 ** 7) < 1e-4


@pytest.mark.parametrize("name", ["orb_a", "orb_b", "orb_gauss_rounded"])
def test_oracle_matches_golden(oracle, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    nf, ini, mn = [int(v) for v in g["params"]]
    o = oracle.OrbOracle(nfeatures=nf, ini_th_fast=ini, min_th_fast=mn)
    oracle.set_gauss_variant("rounded" if name == "orb_gauss_rounded" else "error-diffused")
    try:
        mono, kps, desc = o.extract(g["image"])
        blurred3 = o.blurred(3)
    finally:
        oracle.set_gauss_variant("error-diffused")
    assert mono == int(g["mono"])
    kp = g["keypoints"]
    for i, f in enumerate(("x", "y", "size", "angle", "response")):
        assert np.array_equal(kps[f], kp[:, i]), f
    assert np.array_equal(kps["octave"], kp[:, 5].astype(np.int32))
    assert np.array_equal(desc, g["descriptors"])
    assert np.array_equal(o.level(7), g["level7"])
    assert np.array_equal(blurred3, g["blurred3"])


def test_orb_structure(oracle, synthetic):
    left, _ = synthetic.stereo_pair(5, 640, 240)
    o = oracle.OrbOracle(nfeatures=1000)
    mono, kps, desc = o.extract(left)
    assert mono == len(kps) and 900 < len(kps) <= 1000 + 4 * 8
    assert np.all(np.diff(kps["octave"]) >= 0)  # level-major order (ORBextractor.cc:1095)
    s = o.tables()[0]
    assert np.all(kps["size"] == np.floor(31 * s[kps["octave"]]))
    assert np.all((kps["angle"] >= 0) & (kps["angle"] < 360))
    assert kps["x"].min() >= 19 and kps["x"].max() < 640 - 19 + 4
    # descriptors are not degenerate
    bits = np.unpackbits(desc, axis=1).mean()
    assert 0.3 < bits < 0.7
    # empty image -> -1 (ORBextractor.cc:1063-1064)
    assert o.extract(np.zeros((0, 0), np.uint8))[0] == -1
    # a flat image has no corners at all
    assert len(o.extract(np.full((240, 320), 90, np.uint8))[1]) == 0


def test_gaussian_variants(oracle, golden_dir):
    """The two candidate 8.8 kernels of cv::GaussianBlur(7 x 7, sigma 2) on 8-bit images (which one OpenCV 4.2 uses cannot be checked here):
    both follow from exp(-x^2 / 8) / sum scaled by 256 -- per-tap rounding gives {18, 34, 49, 55} (sum 257), spreading the rounding error
    so that the sum stays 256 gives {18, 34, 48, 56}.  The second is (257 / 256)^2 brighter: up to two grey levels; keypoints do not depend on the blur."""
    w = np.exp(-np.arange(-3, 4) ** 2 / 8.0)
    w = 256 * w / w.sum()
    assert np.rint(w).astype(int).tolist() == [18, 34, 49, 55, 49, 34, 18]
    err, ed = 0.0, []
    for v in w[:4]:  # error diffusion from the border towards the centre
        q = np.floor(v + err + 0.5)
        err += v - q
        ed.append(int(q))
    assert ed == [18, 34, 48, 56] and 2 * sum(ed[:3]) + ed[3] == 256
    a, r = np.load(os.path.join(golden_dir, "orb_a.npz")), np.load(os.path.join(golden_dir, "orb_gauss_rounded.npz"))
    assert np.array_equal(a["keypoints"], r["keypoints"]) and np.array_equal(a["level7"], r["level7"])
    d = a["blurred3"].astype(int) - r["blurred3"].astype(int)
    assert 1 <= np.abs(d).max() <= 2 and (d != 0).mean() > 0.05 and d.mean() < 0
    bits = np.unpackbits(a["descriptors"] ^ r["descriptors"], axis=1).sum(1)
    assert 0 < bits.mean() < 12  # a few of the 256 comparisons flip per descriptor
    # float64 separable Gaussian on the same image: each variant stays within one grey level of it
    img = a["image"].astype(np.float64)
    for name in ("error-diffused", "rounded"):
        oracle.set_gauss_variant(name)
        try:
            got = oracle.gaussian_blur7(a["image"]).astype(np.float64)
        finally:
            oracle.set_gauss_variant("error-diffused")
        k = np.exp(-np.arange(-3, 4) ** 2 / 8.0); k /= k.sum()
        pad = np.pad(img, 3, mode="reflect")
        hz = sum(k[i] * pad[:, i:i + img.shape[1]] for i in range(7))
        ref = sum(k[i] * hz[i:i + img.shape[0]] for i in range(7))
        assert np.abs(got - ref).max() <= (1.0 if name == "error-diffused" else 2.6)


def test_torch_crosscheck(oracle, golden_dir):
    """The oracle's fixed-point restatements against PyTorch's float operators on the same inputs (tools/make_golden_torch_crosscheck.py):
    resize within one grey level (11-bit weights, two roundings), blur within one grey level, fastAtan2 within 0.3 degrees."""
    g = np.load(os.path.join(golden_dir, "torch_crosscheck.npz"))
    img = g["image"]
    for k in range(3):
        w, h = [int(v) for v in g["resize_size_%d" % k]]
        got = oracle.resize_linear(img, w, h).astype(np.float64)
        ref = g["resize_%d" % k].astype(np.float64)
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 1.0 and np.abs(got - ref).mean() < 0.3
    got = oracle.gaussian_blur7(img).astype(np.float64)
    assert np.abs(got - g["blur"]).max() <= 1.0 and np.abs(got - g["blur"]).mean() < 0.3
    a = np.array([oracle.fast_atan2(float(y), float(x)) for y, x in zip(g["atan_y"], g["atan_x"])])
    d = np.abs(a - g["atan_deg"])
    d = np.minimum(d, 360 - d)
    assert d.max() < 0.3
