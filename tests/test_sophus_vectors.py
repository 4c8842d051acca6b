"""The only test inputs the reference tree holds near the hot path: the element / point sets of Sophus' LieGroupTests
(SF/Thirdparty/Sophus/test/core/tests.hpp:508-544, test_se3.cpp:30-50, test_so3.cpp:29-62; data in tests/golden/sophus_inputs.npz, made by
tools/make_golden_sophus_inputs.py).  The identities of tests.hpp that hold for Sophus::SE3f are run on the oracle's float restatements
(oracle/lidar_pose.cpp: inverse, product, log, exp, group action) and on the product's host code (csrc/lidar_pose_host.cpp behind
tc2li_se3_interpolate / tc2li_lidar_update_pose / tc2li_lidar_sync_transform / tc2li_transform_point_cloud).  This pins the SE3f / SO3f
conventions rows b4 and a11 rest on (quaternion order, left / right products, exp o log); it does not lift "parity unpinned" for the hot path.
Tolerance: Sophus' own sqrt(epsilon) for float."""
import os

import numpy as np
import pytest

SQRT_EPS = float(np.sqrt(np.finfo(np.float32).eps))


def mat(p7):
    from scipy.spatial.transform import Rotation
    M = np.eye(4)
    M[:3, :3] = Rotation.from_quat(np.asarray(p7[:4], float)).as_matrix()
    M[:3, 3] = p7[4:]
    return M


def approx(A, B, scale=1.0):
    return np.abs(A - B).max() <= SQRT_EPS * max(1.0, scale) * 4


@pytest.fixture(scope="module")
def vec(golden_dir):
    z = np.load(os.path.join(golden_dir, "sophus_inputs.npz"))
    elems = np.concatenate([z["se3"], z["so3"]]).astype(np.float32)
    return elems, z["points"].astype(np.float32)


def test_group_identities_on_the_oracle(oracle, vec):
    elems, points = vec
    for a in elems:
        A = mat(a)
        s = np.abs(a[4:]).max()
        for b in elems:
            o = oracle.se3f_ops(a, b, 0.5)
            B = mat(b)
            # product and inverse (tests.hpp: groupActionTest / productTest style)
            assert approx(mat(o["mul"]), A @ B, s + np.abs(b[4:]).max())
            assert approx(mat(o["inverse"]) @ A, np.eye(4), s)
        # exp o log (tests.hpp expLogTest)
        assert approx(mat(o["exp_log"]), A, s)
        # log of a unit quaternion rotation: |omega| <= pi
        assert np.linalg.norm(o["log"][3:]) <= np.pi + 1e-3
        # group action on the points (tests.hpp groupActionTest)
        P = np.zeros(len(points), oracle.POINT_DTYPE)
        P["x"], P["y"], P["z"] = points[:, 0], points[:, 1], points[:, 2]
        Q = oracle.transform_point_cloud(P, a)
        want = points.astype(np.float64) @ A[:3, :3].T + A[:3, 3]
        assert approx(np.stack([Q["x"], Q["y"], Q["z"]], 1), want, s)


def test_interpolate_end_points_and_left_invariance(oracle, pkg, vec):
    """InterpolateSE3 (SF/src/Tracking.cc:1552-1563: quaternion slerp + linear translation).  Of Sophus' interpolateAndMeanTest the boundary
    conditions and the left-invariance hold for this form (the right- and inverse-invariance are properties of exp(alpha log(a^-1 b)) only);
    pairs with a shortest-path ambiguity (rotation angle of a^-1 b near pi) are skipped as Sophus does."""
    elems, _ = vec
    for a in elems:
        for b in elems:
            for alpha, want in ((0.0, a), (1.0, b)):
                got = pkg.capi.se3_interpolate(a, b, alpha)
                assert np.array_equal(got, oracle.se3f_ops(a, b, alpha)["interpolate"])   # product == oracle, bit for bit
                assert approx(mat(got), mat(want), np.abs(want[4:]).max())
            rel = np.linalg.inv(mat(a)) @ mat(b)
            ang = np.arccos(np.clip((np.trace(rel[:3, :3]) - 1) / 2, -1, 1))
            if abs(ang - np.pi) < 1e-2:
                continue
            for alpha in (0.1, 0.5, 0.75, 0.99):
                q = pkg.capi.se3_interpolate(a, b, alpha)
                assert np.array_equal(q, oracle.se3f_ops(a, b, alpha)["interpolate"])
                for d in elems[:6]:
                    da, db = oracle.se3f_ops(d, a, 0)["mul"], oracle.se3f_ops(d, b, 0)["mul"]
                    lhs = mat(pkg.capi.se3_interpolate(da, db, alpha))
                    rhs = mat(d) @ mat(q)
                    assert approx(lhs, rhs, np.abs(rhs[:3, 3]).max())


def test_update_lidar_pose_is_exp_of_log_on_the_sophus_set(oracle, pkg, vec, synthetic):
    """UpdateLidarPose (LidarFrontEnd.cpp:786-800): Twc = Tcw_last^-1 * exp(t * log(velocity^-1)).  At t = 1 the exponential undoes the logarithm:
    the product's host code must give Rw2_w1 * Tcw_last^-1 * velocity^-1 * Tcl, formed here from the matrices without exp / log."""
    elems, _ = vec
    Rw = np.array([[0, 0, 1.0], [-1, 0, 0], [0, -1, 0]])
    st0 = np.concatenate([np.eye(3).ravel(), np.zeros(3), np.eye(3).ravel(), np.zeros(3)])
    Tcl = mat(synthetic.TCL7)
    for last in elems[:5]:
        for vel in elems:
            got, _ = pkg.capi.lidar_update_pose(last, vel, 1.0, synthetic.TCL7, st0)
            want, _ = oracle.update_lidar_pose(last, vel, 1.0, synthetic.TCL7, st0)
            assert np.array_equal(got, want)                      # product == oracle
            Twl = np.linalg.inv(mat(last)) @ np.linalg.inv(mat(vel)) @ Tcl
            s = max(1.0, np.abs(Twl[:3, 3]).max())
            assert np.abs(got[:9].reshape(3, 3) - Rw @ Twl[:3, :3]).max() < 8 * SQRT_EPS
            assert np.abs(got[9:12] - Rw @ Twl[:3, 3]).max() < 8 * SQRT_EPS * s
