"""GPU parity of the optimisation back end with the oracle.  Bar (BASELINE.json): optimised SE3 poses within 1e-4
relative; here the inlier/outlier sets must also be identical and the poses agree to ~1e-9 (the only difference is
the order of the floating-point reductions)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POSE_RTOL = 1e-4  # the tolerance BASELINE.json states for optimised SE3 poses


def frame_problem(w, k, truth_points=True):
    ed = w["edges"][w["edges"][:, 1] == k].copy()
    Xw = (w["points_true"] if truth_points else w["points"])[ed[:, 0].astype(int)]
    ed[:, 0] = np.arange(len(ed)); ed[:, 1] = 0
    return Xw, ed


@pytest.mark.parametrize("seed,outliers", [(2, 0.08), (5, 0.0), (7, 0.3)])
def test_pose_optimization(pkg, oracle, synthetic, seed, outliers):
    w = synthetic.ba_window(seed, n_opt=6, n_fix=6, n_points=1500, outlier_frac=outliers)
    for k in (len(w["poses"]) - 1, len(w["poses"]) - 3):
        Xw, ed = frame_problem(w, k)
        want_pose, want_out, want_inl, _ = oracle.pose_optimization(w["poses"][k], Xw, ed, w["cam"])
        pose, out, inl = pkg.pose_optimization(w["poses"][k], Xw, pkg.pack_ba_edges(ed), w["cam"])
        assert inl == want_inl
        assert np.array_equal(out, want_out)
        assert np.allclose(pose, want_pose, rtol=POSE_RTOL, atol=1e-7)
        assert np.abs(pose - want_pose).max() < 1e-6  # in practice identical up to the float rounding of SetPose


def test_pose_optimization_edge_cases(pkg, oracle, synthetic):
    w = synthetic.ba_window(3, n_opt=3, n_fix=3, n_points=400)
    k = len(w["poses"]) - 1
    Xw, ed = frame_problem(w, k)
    for n in (0, 2, 3, 9, 10, 40):
        want_pose, want_out, want_inl, _ = oracle.pose_optimization(w["poses"][k], Xw[:n], ed[:n], w["cam"])
        pose, out, inl = pkg.pose_optimization(w["poses"][k], Xw[:n], pkg.pack_ba_edges(ed[:n]), w["cam"])
        assert inl == want_inl and np.array_equal(out, want_out)
        assert np.allclose(pose, want_pose, rtol=POSE_RTOL, atol=1e-7)
    # monocular-only correspondences
    edm = ed.copy(); edm[:, 4] = -1
    want_pose, want_out, want_inl, _ = oracle.pose_optimization(w["poses"][k], Xw, edm, w["cam"])
    pose, out, inl = pkg.pose_optimization(w["poses"][k], Xw, pkg.pack_ba_edges(edm), w["cam"])
    assert inl == want_inl and np.array_equal(out, want_out) and np.allclose(pose, want_pose, rtol=POSE_RTOL, atol=1e-7)


def test_pose_optimization_batch(pkg, oracle, synthetic):
    w = synthetic.ba_window(4, n_opt=8, n_fix=4, n_points=1200, outlier_frac=0.1)
    ks = list(range(4, 12))
    probs = [frame_problem(w, k) for k in ks]
    offs = np.concatenate([[0], np.cumsum([len(e) for _, e in probs])])
    Xw = np.concatenate([x for x, _ in probs])
    ed = np.concatenate([e for _, e in probs])
    poses, out, inl = pkg.pose_optimization_batch(w["poses"][ks], offs, Xw, pkg.pack_ba_edges(ed), w["cam"])
    for i, k in enumerate(ks):
        want_pose, want_out, want_inl, _ = oracle.pose_optimization(w["poses"][k], probs[i][0], probs[i][1], w["cam"])
        assert inl[i] == want_inl
        assert np.array_equal(out[offs[i]:offs[i + 1]], want_out)
        assert np.allclose(poses[i], want_pose, rtol=POSE_RTOL, atol=1e-7)
