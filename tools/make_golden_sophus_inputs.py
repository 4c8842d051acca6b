"""tests/golden/sophus_inputs.npz: the INPUT sets of Sophus' own Lie-group unit tests -- the only test vectors the reference tree holds near
this path (SF/Thirdparty/Sophus/test/core/tests.hpp:508-544 getTestSE3s, test_se3.cpp:30-50 tangents and points, test_so3.cpp:29-62) -- as
data: SE3 elements as unit quaternion (x y z w) + translation, composed here with scipy in double precision exactly as those files spell
them.  tests/test_sophus_vectors.py runs the LieGroupTests identities (tests.hpp) that apply to Sophus::SE3f on the oracle's and the
product's restatements of exp / log / inverse / products / InterpolateSE3.  Data only: no Sophus source text."""
import os
import numpy as np
from scipy.spatial.transform import Rotation as R

PI = np.pi


def se3(rotvec=(0, 0, 0), t=(0, 0, 0)):
    return R.from_rotvec(rotvec).as_matrix(), np.asarray(t, float)


def mul(a, b):
    return a[0] @ b[0], a[0] @ b[1] + a[1]


def rot(axis, ang):
    v = np.zeros(3); v["xyz".index(axis)] = ang
    return se3(v)


se3s = [
    se3((0.2, 0.5, 0.0)),
    se3((0.2, 0.5, -1.0), (10, 0, 0)),
    se3(t=(0, 100, 5)),
    rot("z", 0.00001),
    mul(se3(t=(0, -0.00000001, 0.0000000001)), rot("z", 0.00001)),
    mul(se3(t=(0.01, 0, 0)), rot("z", 0.00001)),
    mul(se3(t=(4, -5, 0)), rot("x", PI)),
    mul(mul(se3((0.2, 0.5, 0.0)), rot("x", PI)), se3((-0.2, -0.5, -0.0))),
    mul(mul(se3((0.3, 0.5, 0.1), (2, 0, -7)), rot("x", PI)), se3((-0.3, -0.5, -0.1), (0, 6, 0))),
]
se3_tangents = np.array([[0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0], [0, 1, 0, 1, 0, 0], [0, -5, 10, 0, 0, 0], [-1, 1, 0, 0, 0, 1], [20, -1, 0, -1, 1, 0],
                         [30, 5, -1, 20, -1, 0]], float)  # upsilon (3), omega (3)
points = np.array([[1, 2, 4], [1, -3, 0.5]], float)
so3s = [
    R.from_quat([0.0, 1.0, 0.0, 0.1e-11]).as_matrix(),   # Eigen::Quaternion(w = 0.1e-11, x = 0, y = 1, z = 0), normalised by SO3's constructor
    R.from_quat([0.00001, 0.0, 0.0, -1.0]).as_matrix(),
    se3((0.2, 0.5, 0.0))[0], se3((0.2, 0.5, -1.0))[0], np.eye(3), se3((0, 0, 0.00001))[0], se3((PI, 0, 0))[0],
    se3((0.2, 0.5, 0.0))[0] @ se3((PI, 0, 0))[0] @ se3((-0.2, -0.5, -0.0))[0],
    se3((0.3, 0.5, 0.1))[0] @ se3((PI, 0, 0))[0] @ se3((-0.3, -0.5, -0.1))[0],
]
so3_tangents = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [PI / 2, PI / 2, 0], [-1, 1, 0], [20, -1, 0], [30, 5, -1]], float)


def q7(Rm, t):
    q = R.from_matrix(Rm).as_quat()  # x y z w
    return np.concatenate([q, t])


out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sophus_inputs.npz")
np.savez(out, se3=np.array([q7(*T) for T in se3s]), se3_tangents=se3_tangents, points=points,
         so3=np.array([q7(Rm, np.zeros(3)) for Rm in so3s]), so3_tangents=so3_tangents)
print("wrote", out)
