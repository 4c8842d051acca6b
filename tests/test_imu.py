"""IMU pre-integration (host-only, no GPU): the product's host implementation (float, through the
C ABI) against the oracle (double evaluation of the same recursions) and against closed forms.  Tolerances: the reference
computes in float, so agreement is ~1e-5 relative."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation


NOISE = None


@pytest.fixture(scope="module")
def noise(synthetic):
    return synthetic.IMU_NOISE


def close(a, b, rtol=2e-5, atol=2e-6):
    return np.allclose(a, b, rtol=rtol, atol=atol * max(1.0, float(np.abs(b).max())))


@pytest.mark.parametrize("seed,t0,t1,jitter", [(0, 1.003, 1.104, 0.0), (1, 2.5, 2.6, 0.0), (2, 0.101, 0.499, 0.3), (3, 5.0, 5.021, 0.0)])
def test_preintegration_matches_the_oracle(pkg, oracle, synthetic, noise, seed, t0, t1, jitter):
    s = synthetic.imu_samples(t0, t1, seed=seed, jitter=jitter)
    bias = np.array([0.02, -0.01, 0.03, 0.001, -0.002, 0.0005], np.float32)
    p = pkg.capi.Preintegrated(bias, *noise)
    steps = p.preintegrate(s, t0, t1)
    osteps, want = oracle.imu_preintegrate(s, t0, t1, bias, *noise)
    got = p.fields()
    assert steps == osteps == len(s) - 1
    assert abs(got["dT"] - want["dT"]) < 1e-6 and abs(got["dT"] - (t1 - t0)) < 1e-5
    for name in ("dR", "dV", "dP", "JRg", "JVg", "JVa", "JPg", "JPa", "avgA", "avgW"):
        assert close(got[name], want[name]), name
    assert np.allclose(got["C"], want["C"], rtol=1e-3, atol=1e-6 * np.abs(want["C"]).max())
    # covariance: symmetric positive semi-definite, random-walk block = steps * NgaWalk
    C = got["C"].astype(np.float64)
    assert np.allclose(C, C.T, atol=1e-6 * np.abs(C).max())
    assert np.linalg.eigvalsh(0.5 * (C[:9, :9] + C[:9, :9].T)).min() > -1e-6 * np.abs(C).max()
    assert np.allclose(np.diag(C)[9:12], steps * noise[2] ** 2, rtol=1e-4) and np.allclose(np.diag(C)[12:], steps * noise[3] ** 2, rtol=1e-4)
    # dR is a rotation
    assert np.allclose(got["dR"] @ got["dR"].T, np.eye(3), atol=1e-6) and abs(np.linalg.det(got["dR"].astype(np.float64)) - 1) < 1e-6


@pytest.mark.parametrize("seed,t0,t1,jitter", [(0, 1.003, 1.104, 0.0), (1, 2.5, 2.6, 0.0), (2, 0.101, 0.499, 0.3), (3, 5.0, 5.021, 0.0), (7, 3.0, 4.0, 0.5)])
def test_preintegration_equals_the_float_oracle_bit_for_bit(pkg, oracle, synthetic, noise, seed, t0, t1, jitter):
    """the oracle's FLOAT evaluation (IntegrateNewMeasurementFloat: the reference's Eigen expressions one node at a
    time in the order the source associates them, ImuTypes.cc:186-244, NormalizeRotation as Eigen's two-sided Jacobi SVD) against
    tc2li_imu_preintegrate: every field of IMU::Preintegrated, the covariance included, bit for bit -- the order of evaluation is now
    tested, not only the value to 2e-5 (the double form above stays as the accuracy check)."""
    s = synthetic.imu_samples(t0, t1, seed=seed, jitter=jitter)
    bias = np.array([0.02, -0.01, 0.03, 0.001, -0.002, 0.0005], np.float32)
    p = pkg.capi.Preintegrated(bias, *noise)
    steps = p.preintegrate(s, t0, t1)
    osteps, want = oracle.imu_preintegrate(s, t0, t1, bias, *noise, float_eval=True)
    got = p.fields()
    assert steps == osteps and np.float32(got["dT"]) == np.float32(want["dT"])
    for name in ("dR", "dV", "dP", "JRg", "JVg", "JVa", "JPg", "JPa", "avgA", "avgW", "C"):
        assert np.array_equal(got[name], want[name]), (name, np.abs(got[name] - want[name]).max())
    # and the float evaluation is the double one to float rounding
    _, dbl = oracle.imu_preintegrate(s, t0, t1, bias, *noise)
    for name in ("dR", "dV", "dP", "JRg", "JVg", "JVa", "JPg", "JPa"):
        assert close(want[name], dbl[name]), name


def test_normalize_rotation_float_is_a_rotation_close_to_the_polar_factor(oracle):
    """The float SVD form of IMU::NormalizeRotation (restated from Eigen's JacobiSVD) against the double polar factor: orthogonal, det +1,
    the same matrix to float rounding -- for near-rotations (what the pre-integration feeds it) and for scaled / skewed inputs."""
    rng = np.random.default_rng(3)
    for k in range(200):
        R = Rotation.from_rotvec(rng.normal(0, 1.0, 3)).as_matrix()
        if k % 2:
            R = R + rng.normal(0, 1e-3 if k % 4 == 1 else 5e-2, (3, 3))
        if k % 5 == 0:
            R = R * rng.uniform(0.5, 20.0)
        got = oracle.normalize_rotation(R.astype(np.float32), float_eval=True).astype(np.float64)
        want = oracle.normalize_rotation(R.astype(np.float32)).astype(np.float64)
        assert np.allclose(got @ got.T, np.eye(3), atol=5e-6) and abs(np.linalg.det(got) - 1) < 5e-6
        assert np.allclose(got, want, atol=3e-6), (k, np.abs(got - want).max())


def test_constant_signal_closed_form(pkg, noise):
    """Constant angular velocity about z and constant specific force: dR = Exp(w T); dV, dP against a fine numerical integral."""
    w = np.array([0, 0, 0.3], np.float32); a = np.array([1.0, 0.5, 9.0], np.float32)
    p = pkg.capi.Preintegrated(np.zeros(6, np.float32), *noise)
    dt, n = 0.01, 50
    for _ in range(n):
        p.IntegrateNewMeasurement(a, w, dt)
    f = p.fields()
    T = n * dt
    assert np.allclose(f["dR"], Rotation.from_rotvec(w * T).as_matrix(), atol=2e-6)
    # the reference's scheme holds R fixed within a step: compare with the same discrete sum in float64
    R = np.eye(3); V = np.zeros(3); P = np.zeros(3)
    for _ in range(n):
        P = P + V * dt + 0.5 * R @ a * dt * dt
        V = V + R @ a * dt
        R = R @ Rotation.from_rotvec(w.astype(np.float64) * dt).as_matrix()
    assert np.allclose(f["dV"], V, rtol=1e-5, atol=1e-6) and np.allclose(f["dP"], P, rtol=1e-5, atol=1e-6)
    assert np.allclose(f["avgW"], w, atol=1e-6)


def test_bias_jacobians_against_reintegration(pkg, synthetic, noise):
    """First-order bias correction (GetDelta*) against re-integrating with the new bias."""
    s = synthetic.imu_samples(3.0, 3.4, seed=5, noise=False)
    b0 = np.zeros(6, np.float32)
    p = pkg.capi.Preintegrated(b0, *noise)
    p.preintegrate(s, 3.0, 3.4)
    for k in range(6):
        b1 = b0.copy(); b1[k] = 0.01
        dR, dV, dP = p.delta(b1)
        q = pkg.capi.Preintegrated(b1, *noise)
        q.preintegrate(s, 3.0, 3.4)
        f = q.fields()
        assert np.allclose(dR, f["dR"], atol=3e-5), k
        assert np.allclose(dV, f["dV"], atol=3e-4) and np.allclose(dP, f["dP"], atol=1e-4), k
    # and the original bias reproduces the stored deltas
    dR, dV, dP = p.delta(b0)
    f = p.fields()
    assert np.allclose(dR, f["dR"], atol=1e-6) and np.array_equal(dV, f["dV"]) and np.array_equal(dP, f["dP"])


def test_predict_state(pkg, oracle, synthetic, noise):
    s = synthetic.imu_samples(7.0, 7.1, seed=9)
    bias = np.array([0.01, 0.0, -0.02, 0.002, 0.001, -0.001], np.float32)
    bias_kf = bias + np.array([0.003, -0.002, 0.001, 1e-4, -2e-4, 1e-4], np.float32)
    Rwb1 = Rotation.from_rotvec([0.1, -0.4, 0.2]).as_matrix().astype(np.float32)
    twb1, Vwb1 = np.array([3.0, -1.0, 0.5], np.float32), np.array([8.0, 0.3, -0.1], np.float32)
    p = pkg.capi.Preintegrated(bias, *noise)
    p.preintegrate(s, 7.0, 7.1)
    R2, t2, v2 = p.predict_state(bias_kf, Rwb1, twb1, Vwb1)
    want = oracle.imu_predict(s, 7.0, 7.1, bias, bias_kf, *noise, Rwb1, twb1, Vwb1)
    assert close(R2, want[0]) and close(t2, want[1]) and close(v2, want[2])
    dR, dV, dP = p.delta(bias_kf)
    assert close(dR, want[3]) and close(dV, want[4]) and close(dP, want[5])
    # physics: gravity along -z of the world frame
    T = p.fields()["dT"]
    assert np.allclose(v2, Vwb1 + T * np.array([0, 0, -9.81]) + Rwb1 @ dV, atol=1e-5)


def test_normalize_rotation_is_the_polar_factor(pkg, oracle):
    rng = np.random.default_rng(3)
    for _ in range(20):
        R = Rotation.from_rotvec(rng.normal(0, 1, 3)).as_matrix() + rng.normal(0, 1e-3, (3, 3))
        U, _, Vt = np.linalg.svd(R)
        assert np.allclose(oracle.normalize_rotation(R), U @ Vt, atol=1e-6)


def test_edge_cases(pkg, noise):
    p = pkg.capi.Preintegrated(np.zeros(6, np.float32), *noise)
    assert p.preintegrate(np.zeros(0, pkg.capi.IMU_SAMPLE_DTYPE), 0.0, 0.1) == 0      # no samples
    assert p.preintegrate(np.zeros(1, pkg.capi.IMU_SAMPLE_DTYPE), 0.0, 0.1) == 0      # "Empty IMU measurements vector"
    s = np.zeros(2, pkg.capi.IMU_SAMPLE_DTYPE); s["t"] = [0.0, 0.1]; s["a"][:, 2] = 9.81
    assert p.preintegrate(s, 0.02, 0.08) == 1                                            # single step spans the whole interval
    f = p.fields()
    assert abs(f["dT"] - 0.06) < 1e-6 and np.allclose(f["dV"], [0, 0, 9.81 * 0.06], atol=1e-5)
    # tiny rotation takes the first-order branch
    p2 = pkg.capi.Preintegrated(np.zeros(6, np.float32), *noise)
    p2.IntegrateNewMeasurement(np.zeros(3, np.float32), np.array([1e-4, 0, 0], np.float32), 0.01)
    assert np.allclose(p2.fields()["dR"], np.eye(3), atol=2e-6)
