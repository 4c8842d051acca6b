"""Plane extraction of the LiDAR window on the device (balm_cut_kernels.hip; cut_voxel + recut + push_voxel, SF/src/bavoxel.cc:42-91,
SF/include/bavoxel.h:57-78,492-602): the kernels' clusters against the host restatement the library keeps for the one-window entry
points (tc2li_host_lidar_planes, itself pinned on the oracle in test_oracle_balm.py) -- the same planes in the same order, every sum
bit for bit -- and against the oracle directly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _window(synthetic, seed, W, n_pts, pose_noise=(0.1, 0.01)):
    w = synthetic.ba_window(seed, n_opt=max(6, W), n_fix=4, n_points=200, pose_noise=pose_noise)
    last = len(w["poses"]) - 1
    win = list(range(last, last - W, -1))
    return w, win, synthetic.ba_window_clouds(w, win, n_points=n_pts, seed=seed)


@pytest.mark.parametrize("seed,W,n_pts", [(0, 6, 3000), (1, 4, 2400), (2, 2, 1500), (3, 7, 2000), (4, 3, 9000), (5, 5, 300), (6, 6, 40), (7, 1, 2000)])
def test_device_planes_equal_host_planes(pkg, synthetic, seed, W, n_pts):
    w, win, clouds = _window(synthetic, seed, W, n_pts)
    hc, hcoe = pkg.capi.lidar_planes_host(w["poses"], win, clouds, synthetic.TCL7)
    dc, dcoe, info = pkg.capi.lidar_planes_device(w["poses"], win, clouds, synthetic.TCL7)
    assert info[1] == 0 and info[0] == len(hcoe) == len(dcoe)
    if W >= 2 and n_pts >= 1500:
        assert len(hcoe) > 10
    if W == 1:
        assert len(hcoe) == 0  # a plane must be seen from two keyframes
    assert np.array_equal(dcoe, hcoe)
    assert np.array_equal(dc, hc)  # the planes in the host walk's order, every sum in the host's order


def test_device_planes_against_the_oracle(pkg, oracle, synthetic):
    w, win, clouds = _window(synthetic, 11, 5, 2500)
    oc, ocoe = oracle.lidar_planes(w["poses"], win, clouds, synthetic.TCL7)
    dc, dcoe, _ = pkg.capi.lidar_planes_device(w["poses"], win, clouds, synthetic.TCL7)
    o10 = np.concatenate([oc[:, :, [0, 1, 2, 4, 5, 8]], oc[:, :, 9:13]], 2)

    def canon(c, q):  # the oracle walks an unordered map: compare as sets
        order = np.lexsort(np.concatenate([c.reshape(len(q), -1).T, q[None]], 0))
        return c[order], q[order]
    a, ac = canon(o10, ocoe)
    b, bc = canon(dc, dcoe)
    assert len(ocoe) > 20 and np.array_equal(ac, bc) and np.array_equal(a, b)


def test_points_on_voxel_faces_and_in_one_voxel(pkg, synthetic):
    """Coordinates on the voxel faces and the octant planes (the float floor of cut_voxel, the > of recut), negative coordinates, a
    dense blob that is split twice."""
    w, win, clouds = _window(synthetic, 21, 4, 1200)
    rng = np.random.default_rng(3)
    extra = []
    for k in range(4):
        grid = rng.integers(-12, 12, (600, 3)).astype(np.float32) * 0.25  # multiples of the octant edge
        blob = (rng.normal(0, 0.12, (900, 3)) + [3.3, -2.6, 0.4]).astype(np.float32)
        plane = np.stack([rng.uniform(-4, 4, 1500), rng.uniform(-4, 4, 1500), np.full(1500, -1.0) + rng.normal(0, 0.004, 1500)], 1).astype(np.float32)
        extra.append(np.concatenate([clouds[k], grid, blob, plane]))
    hc, hcoe = pkg.capi.lidar_planes_host(w["poses"], win, extra, synthetic.TCL7)
    dc, dcoe, info = pkg.capi.lidar_planes_device(w["poses"], win, extra, synthetic.TCL7)
    assert info[1] == 0 and len(hcoe) > 20
    assert np.array_equal(dcoe, hcoe) and np.array_equal(dc, hc)


def test_a_window_outside_the_kernels_range_is_declined(pkg, synthetic):
    w, win, clouds = _window(synthetic, 31, 3, 500)
    far = [c.copy() for c in clouds]
    far[1][7] = [3.0e6, 0, 0]  # beyond the key fields
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.lidar_planes_device(w["poses"], win, far, synthetic.TCL7)
    w8, win8, clouds8 = _window(synthetic, 32, 8, 300)
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.lidar_planes_device(w8["poses"], win8, clouds8, synthetic.TCL7)


def test_batch_takes_the_host_extraction_for_a_declined_window(pkg, synthetic):
    """A lock-step batch whose second window has a point beyond the key range: that window's planes come from the host, the result of
    every window is the one-window entry point's."""
    windows, singles = [], []
    for seed in range(3):
        w = synthetic.ba_window(40 + seed, n_opt=6, n_fix=8, n_points=600, pose_noise=(0.1, 0.01))
        e = pkg.pack_ba_edges(w["edges"])
        last = len(w["poses"]) - 1
        win = list(range(last, last - 4, -1))
        clouds = synthetic.ba_window_clouds(w, win, n_points=2000)
        if seed == 1:
            clouds[2][5] = [2.5e6, 1.0, 1.0]
        windows.append(dict(poses=w["poses"], fixed=w["fixed"], points=w["points"], edges=e, win_pose=win, clouds=clouds, Tcl7=synthetic.TCL7, weight=1.0))
        singles.append(pkg.capi.local_lv_bundle_adjustment(w["poses"], w["fixed"], w["points"], e, w["cam"], win, clouds, synthetic.TCL7, 1.0))
        cam = w["cam"]
    batch = pkg.capi.BaBatch(windows, cam)
    assert batch.run(max_concurrency=1) == len(windows)
    for i, s in enumerate(singles):
        r = batch.result(i)
        assert r[4].trials == s[4].trials and r[5].n_planes == s[5].n_planes and r[5].residual == s[5].residual, i
        assert np.array_equal(r[0], s[0]) and np.array_equal(r[1], s[1]), i
