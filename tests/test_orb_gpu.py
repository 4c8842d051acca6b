"""GPU parity tests of the HIP ORB extractor against the CPU oracle: bit-exact pyramids, blurred levels, FAST
candidates, keypoints and descriptors (the bar of BASELINE.json: 'bit-exact for ORB descriptors/keypoint
indices').  Everything goes through the C ABI (tc2li_orb_*)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("x", "y", "size", "angle", "response", "octave")


def assert_same_features(got, want):
    gm, gk, gd = got
    wm, wk, wd = want
    assert gm == wm
    assert len(gk) == len(wk)
    for f in FIELDS:
        assert np.array_equal(gk[f], wk[f]), f
    assert np.array_equal(gd, wd)


@pytest.fixture(scope="module")
def ext(pkg):
    assert pkg.device_count() >= 1, "GPU tests need a device; the library has no fallback"
    e = pkg.OrbExtractor(max_width=1242, max_height=376, max_images=4)
    yield e
    e.close()


def test_full_size_stages_bit_exact(ext, oracle, synthetic):
    left, right = synthetic.stereo_pair(0)
    o = oracle.OrbOracle()
    for img in (left, right):
        want = o.extract(img)
        got = ext.extract(img)
        for lvl in range(8):
            assert ext.level_size(lvl) == o.level(lvl).shape[::-1]
            assert np.array_equal(ext.pyramid_level(0, lvl), o.level(lvl)), "pyramid level %d" % lvl
            assert np.array_equal(ext.blurred_level(0, lvl), o.blurred(lvl)), "blurred level %d" % lvl
            assert np.array_equal(ext.candidates(0, lvl), o.candidates(lvl)), "FAST candidates level %d" % lvl
        assert_same_features(got, want)
        assert len(got[1]) >= 2000


def test_tables_match(ext, oracle):
    s, per_level, _ = oracle.OrbOracle().tables()
    assert np.array_equal(ext.GetScaleFactors(), s)
    assert np.array_equal(ext.features_per_level(), per_level)
    assert np.array_equal(ext.GetInverseScaleFactors(), (np.float32(1) / s).astype(np.float32))
    assert ext.GetLevels() == 8


@pytest.mark.parametrize("seed,w,h", [(1, 1241, 376), (2, 1226, 370), (3, 640, 480), (4, 333, 257), (5, 250, 150)])
def test_other_sizes(pkg, oracle, synthetic, seed, w, h):
    left, _ = synthetic.stereo_pair(seed, w, h)
    e = pkg.OrbExtractor(nfeatures=1200, ini_th_fast=12, max_width=w, max_height=h, max_images=1)
    o = oracle.OrbOracle(nfeatures=1200, ini_th_fast=12)
    assert_same_features(e.extract(left), o.extract(left))
    e.close()


def test_strided_input_and_handle_reuse(ext, oracle, synthetic):
    left, right = synthetic.stereo_pair(6)
    wide = np.zeros((375, 1300), np.uint8)
    wide[:, :1242] = right
    view = wide[:, :1242]  # row stride 1300
    o = oracle.OrbOracle()
    assert_same_features(ext.extract(view), o.extract(right))
    # smaller image through the same handle, then the large one again
    small = np.ascontiguousarray(left[40:340, 100:900])
    assert_same_features(ext.extract(small), o.extract(small))
    assert_same_features(ext.extract(left), o.extract(left))


def test_edge_cases(ext, oracle):
    # empty image: -1 like the reference (ORBextractor.cc:1063-1064)
    mono, k, d = ext.extract(np.zeros((0, 0), np.uint8))
    assert mono == -1 and len(k) == 0
    # flat image: no corners, zero keypoints, monoIndex 0
    flat = np.full((375, 1242), 77, np.uint8)
    mono, k, d = ext.extract(flat)
    assert mono == 0 and len(k) == 0
    # low-contrast texture: cells fall back to minThFAST (ORBextractor.cc:816-820)
    rng = np.random.default_rng(3)
    low = (100 + rng.integers(0, 14, (375, 1242))).astype(np.uint8)
    low[::9, ::11] += 16
    o = oracle.OrbOracle()
    want = o.extract(low)
    assert_same_features(ext.extract(low), want)
    # saturated checkerboard: plateaus of equal scores, everything suppressed or tie-broken identically
    yy, xx = np.mgrid[0:375, 0:1242]
    chk = (((yy // 6) + (xx // 6)) % 2 * 255).astype(np.uint8)
    assert_same_features(ext.extract(chk), o.extract(chk))
    # pure noise: maximal candidate load
    noise = rng.integers(0, 256, (375, 1242)).astype(np.uint8)
    assert_same_features(ext.extract(noise), o.extract(noise))


def test_lapping_area(ext, oracle, synthetic):
    left, _ = synthetic.stereo_pair(8)
    o = oracle.OrbOracle()
    want = o.extract(left, (400, 800))
    got = ext.extract(left, (400, 800))
    assert want[0] < len(want[1])
    assert_same_features(got, want)


@pytest.mark.parametrize("name", ["orb_a", "orb_b"])
def test_golden_vectors(pkg, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    nf, ini, mn = [int(v) for v in g["params"]]
    img = g["image"]
    e = pkg.OrbExtractor(nfeatures=nf, ini_th_fast=ini, min_th_fast=mn, max_width=img.shape[1], max_height=img.shape[0],
                         max_images=1)
    mono, kps, desc = e.extract(img)
    assert mono == int(g["mono"])
    kp = g["keypoints"]
    for i, f in enumerate(FIELDS[:5]):
        assert np.array_equal(kps[f], kp[:, i]), f
    assert np.array_equal(kps["octave"], kp[:, 5].astype(np.int32))
    assert np.array_equal(desc, g["descriptors"])
    assert np.array_equal(e.pyramid_level(0, 7), g["level7"])
    assert np.array_equal(e.blurred_level(0, 3), g["blurred3"])
    e.close()


def test_gaussian_rounded_taps_switch(pkg, golden_dir, monkeypatch):
    """TC2LI_GAUSS_TAPS=rounded at creation selects the per-tap rounded 8.8 Gaussian: the product then reproduces the oracle's second
    golden file (descriptors and blurred level bit for bit); without it, the first."""
    g = np.load(os.path.join(golden_dir, "orb_gauss_rounded.npz"))
    nf, ini, mn = [int(v) for v in g["params"]]
    img = g["image"]
    monkeypatch.setenv("TC2LI_GAUSS_TAPS", "rounded")
    e = pkg.OrbExtractor(nfeatures=nf, ini_th_fast=ini, min_th_fast=mn, max_width=img.shape[1], max_height=img.shape[0], max_images=1)
    monkeypatch.delenv("TC2LI_GAUSS_TAPS")
    mono, kps, desc = e.extract(img)
    assert mono == int(g["mono"]) and np.array_equal(desc, g["descriptors"]) and np.array_equal(e.blurred_level(0, 3), g["blurred3"])
    e.close()
    # a saturated image: with taps summing to 257 the result is clamped at 255
    monkeypatch.setenv("TC2LI_GAUSS_TAPS", "rounded")
    e = pkg.OrbExtractor(nfeatures=100, max_width=320, max_height=200, max_images=1)
    monkeypatch.delenv("TC2LI_GAUSS_TAPS")
    e.extract(np.full((200, 320), 255, np.uint8))
    assert np.all(e.blurred_level(0, 0) == 255)
    e.close()


def test_batch_device_resident(pkg, oracle, synthetic):
    """Batched entry point on images resident in HBM (torch is only the allocator here)."""
    import torch
    frames = synthetic.stereo_batch(3, seed=20)  # [3, 2, H, W]
    n = frames.shape[0] * 2
    h, w = frames.shape[2:]
    dev = torch.from_numpy(frames.reshape(n, h, w)).cuda()
    e = pkg.OrbExtractor(max_width=w, max_height=h, max_images=n)
    kps, desc, counts, mono = e.extract_batch_dev(dev.data_ptr(), n, w, h, w, w * h,
                                                  stream=torch.cuda.current_stream().cuda_stream)
    o = oracle.OrbOracle()
    for i in range(n):
        wm, wk, wd = o.extract(frames.reshape(n, h, w)[i])
        assert counts[i] == len(wk) and mono[i] == wm
        for f in FIELDS:
            assert np.array_equal(kps[i, :counts[i]][f], wk[f]), (i, f)
        assert np.array_equal(desc[i, :counts[i]], wd)
    # size-independent properties at full size: descriptors of a repeated image are identical; order is level-major
    assert np.all(np.diff(kps[0, :counts[0]]["octave"]) >= 0)
    t = e.last_timings()
    assert np.all(t >= 0)
    e.close()


@pytest.mark.parametrize("width", [253, 256, 257, 258, 259, 260, 261, 263, 511, 512, 513, 515, 770])
def test_blur_row_ends(pkg, oracle, synthetic, width):
    """The streaming blur owns 256-column strips: every position of the row end relative to a strip / dword boundary, on the
    caller's image (any pitch) and on the pyramid levels."""
    left = synthetic.stereo_pair(3, 1242, 375)[0]
    img = np.ascontiguousarray(left[60:300, 100:100 + width])  # 240 rows: the smallest level still has a row of FAST cells
    ext = pkg.OrbExtractor(nfeatures=300, max_width=width, max_height=240, max_images=1)
    ora = oracle.OrbOracle(nfeatures=300)
    ext.extract(img)
    ora.extract(img)
    for level in range(8):
        assert np.array_equal(ext.blurred_level(0, level), ora.blurred(level)), (width, level)


@pytest.mark.parametrize("n_images,chunks", [(17, 1), (17, 2), (66, 4), (40, 1)])  # (40, 1): a chunk of 32+ images takes the batch forms (resize: 16 rows per wavefront)
def test_chunked_batch_equals_single_image_calls(pkg, oracle, synthetic, n_images, chunks, monkeypatch):
    """tc2li_orb_extract_batch queues a batch at once, or (TC2LI_ORB_CHUNKS) in chunks of images: every image's features, its
    diagnostics and the stereo matcher's view of the batch are those of single-image calls."""
    monkeypatch.setenv("TC2LI_ORB_CHUNKS", str(chunks))
    import torch
    w, h = 640, 300
    base_l, base_r = synthetic.stereo_pair(9, 1242, 375)
    imgs = np.empty((n_images, h, w), np.uint8)
    for i in range(n_images):  # different windows of the scene, every image distinct
        src = base_l if i % 2 == 0 else base_r
        x0, y0 = (37 * i) % (1242 - w), (11 * i) % (375 - h)
        imgs[i] = src[y0:y0 + h, x0:x0 + w]
    e = pkg.OrbExtractor(max_width=w, max_height=h, max_images=n_images)
    single = pkg.OrbExtractor(max_width=w, max_height=h, max_images=1)
    dev = torch.from_numpy(imgs).cuda()
    kps, desc, counts, mono = e.extract_batch_dev(dev.data_ptr(), n_images, w, h, w, w * h)
    assert e.last_chunks() == chunks
    o = oracle.OrbOracle()
    for i in range(n_images):
        m, k, d = single.extract(imgs[i])
        n = int(counts[i])
        assert n == len(k) and int(mono[i]) == m, i
        for f in FIELDS:
            assert np.array_equal(kps[i, :n][f], k[f]), (i, f)
        assert np.array_equal(desc[i, :n], d), i
    for i in (0, n_images // 2, n_images - 1):  # chunk boundaries against the oracle, with the per-image diagnostics
        want = o.extract(imgs[i])
        assert int(counts[i]) == len(want[1]) and np.array_equal(desc[i, :len(want[1])], want[2])
        assert np.array_equal(e.candidates(i, 0), o.candidates(0)) and np.array_equal(e.blurred_level(i, 3), o.blurred(3))
    # the device-resident features feed the batched stereo matcher: frames (0,1), (2,3), ...
    nf = n_images // 2
    bf, b = float(np.float32(synthetic.BF)), float(np.float32(synthetic.BF) / np.float32(synthetic.FX))
    u_right, depth, _ = pkg.stereo_match_batch(e, nf, bf, b)
    for f in (0, nf - 1):
        ol, orr = oracle.OrbOracle(), oracle.OrbOracle()
        _, kl, dl = ol.extract(imgs[2 * f])
        _, kr, dr = orr.extract(imgs[2 * f + 1])
        wu, wd, _ = oracle.stereo_match(ol, orr, kl, dl, kr, dr, bf, b)
        assert np.array_equal(u_right[f, :len(kl)], wu) and np.array_equal(depth[f, :len(kl)], wd)
    e.close(); single.close()


def test_full_size_batch_forms_equal_single_image_calls(pkg, oracle, synthetic, monkeypatch):
    """A batch of 34 KITTI-sized images takes the forms only batches use -- the pyramid's small levels in one launch (k_resize_tail), four FAST
    cells and four descriptor blocks per workgroup, the keypoint distribution from per-class job lists (272 jobs) -- and gives, image by
    image, what single-image calls give (one workgroup per unit, a launch per level and class); three images also against the oracle.  The
    same batch with those forms switched off is identical too."""
    import torch
    base = synthetic.stereo_batch(4, seed=31)  # [4, 2, H, W]
    h, w = base.shape[2:]
    eight = base.reshape(8, h, w)
    n = 34
    imgs = np.empty((n, h, w), np.uint8)
    for i in range(n):  # distinct images: the eight renderings, shifted by a few columns
        imgs[i] = np.roll(eight[i % 8], 3 * (i // 8), axis=1)
    dev = torch.from_numpy(imgs).cuda()
    e = pkg.OrbExtractor(max_width=w, max_height=h, max_images=n)
    kps, desc, counts, mono = e.extract_batch_dev(dev.data_ptr(), n, w, h, w, w * h)
    kps, desc, counts, mono = kps.copy(), desc.copy(), counts.copy(), mono.copy()
    single = pkg.OrbExtractor(max_width=w, max_height=h, max_images=1)
    for i in range(n):
        m, k, d = single.extract(imgs[i])
        c = int(counts[i])
        assert c == len(k) and int(mono[i]) == m, i
        for f in FIELDS:
            assert np.array_equal(kps[i, :c][f], k[f]), (i, f)
        assert np.array_equal(desc[i, :c], d), i
    o = oracle.OrbOracle()
    for i in (0, 17, 33):
        want = o.extract(imgs[i])
        assert int(counts[i]) == len(want[1]) and np.array_equal(desc[i, :len(want[1])], want[2])
        assert np.array_equal(e.candidates(i, 0), o.candidates(0)) and np.array_equal(e.candidates(i, 5), o.candidates(5))
    e.close()
    # the same batch in a process state where the batch forms are off (the launch-time switches are read per call or per process:
    # the per-call one is the job lists; the others are covered by the single-image comparison above)
    monkeypatch.setenv("TC2LI_QUADTREE_LISTS", "0")
    e2 = pkg.OrbExtractor(max_width=w, max_height=h, max_images=n)
    kps2, desc2, counts2, mono2 = e2.extract_batch_dev(dev.data_ptr(), n, w, h, w, w * h)
    assert np.array_equal(counts2, counts) and np.array_equal(mono2, mono) and np.array_equal(desc2, desc)
    e2.close()
