"""The host's envelope LDL^T of the inertial windows' reduced system (csrc/reduced_solve.hpp; g2o's BlockSolverX over the sparse LinearSolverEigen,
Optimizer.cc:1635-1638) against a dense solve: CPU only."""
import numpy as np
import pytest


def inertial_system(rng, K, lidar=(), gap_link=None, no_imu=()):
    """A symmetric positive definite system with the structure of LocalInertialBA's: 6 pose unknowns per keyframe first, then 9 velocity / bias
    unknowns per keyframe with IMU state; an inertial edge (9 x 24 Jacobian) and two random walks between consecutive keyframes; a dense visual S."""
    imu_of = {}
    for k in range(K):
        if k not in no_imu:
            imu_of[k] = len(imu_of)
    np_, ni = 6 * K, 9 * len(imu_of)
    n = np_ + ni
    Hi = np.zeros((n, n))
    links = [(a, b) for a, b in zip(sorted(imu_of), sorted(imu_of)[1:])]
    if gap_link:
        links.append(gap_link)
    for a, b in links:
        idx = list(range(6 * a, 6 * a + 6)) + list(range(np_ + 9 * imu_of[a], np_ + 9 * imu_of[a] + 9)) + list(range(6 * b, 6 * b + 6)) + \
              list(range(np_ + 9 * imu_of[b], np_ + 9 * imu_of[b] + 3))
        J = rng.normal(0, 1, (9, 24))
        Hi[np.ix_(idx, idx)] += J.T @ J
        for off in (3, 6):
            o1, o2 = np_ + 9 * imu_of[a] + off, np_ + 9 * imu_of[b] + off
            W = np.eye(3) * 40.0
            Hi[o1:o1 + 3, o1:o1 + 3] += W; Hi[o2:o2 + 3, o2:o2 + 3] += W
            Hi[o1:o1 + 3, o2:o2 + 3] -= W; Hi[o2:o2 + 3, o1:o1 + 3] -= W
    if lidar:
        idx = [6 * k + c for k in lidar for c in range(6)]
        A = rng.normal(0, 1, (len(idx), len(idx)))
        Hi[np.ix_(idx, idx)] += A @ A.T
    A = rng.normal(0, 1, (np_, np_))
    S = A @ A.T + 20.0 * np.eye(np_)
    return Hi, S, np_, n


@pytest.mark.parametrize("K,lidar,gap,no_imu", [(25, (19, 20, 21, 22, 23, 24), None, ()), (10, (), None, ()), (3, (0, 1, 2), None, ()), (8, (), (1, 6), ()),
                                              (7, (2, 3), None, (0, 4))])
def test_envelope_ldlt_solves_the_reduced_system(pkg, K, lidar, gap, no_imu):
    rng = np.random.default_rng(K)
    Hi, S, np_, n = inertial_system(rng, K, lidar, gap, no_imu)
    lam = 0.01
    M = Hi.copy()
    M[:np_, :np_] += S
    M[np_:, np_:] += lam * np.eye(n - np_)
    rhs = rng.normal(0, 1, n)
    want = np.linalg.solve(M, rhs)
    got = pkg.capi.host_reduced_solve(Hi, S, lam, rhs)
    assert got is not None
    assert np.abs(got - want).max() <= 1e-9 * max(1.0, np.abs(want).max())
    # the routine reads the lower triangles only
    got2 = pkg.capi.host_reduced_solve(np.tril(Hi), np.tril(S), lam, rhs)
    assert np.array_equal(got, got2)


def test_envelope_ldlt_reports_a_failed_pivot_and_takes_a_window_without_imu_states(pkg):
    rng = np.random.default_rng(1)
    Hi, S, np_, n = inertial_system(rng, 4)
    Hi[np_ + 2, :] = 0; Hi[:, np_ + 2] = 0   # an unknown without any entry: its pivot is lambda = 0
    assert pkg.capi.host_reduced_solve(Hi, S, 0.0, np.ones(n)) is None
    # pose unknowns only (a window none of whose keyframes carries IMU states): the dense pose block
    A = rng.normal(0, 1, (18, 18))
    S2 = A @ A.T + np.eye(18)
    x = pkg.capi.host_reduced_solve(np.zeros((18, 18)), S2, 1.0, np.arange(18.0))
    assert np.allclose(x, np.linalg.solve(S2, np.arange(18.0)), rtol=1e-10, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("K,lidar,no_imu", [(25, (19, 20, 21, 22, 23, 24), ()), (24, (), ()), (10, (4, 5, 6, 7, 8, 9), ()), (3, (), ()), (9, (2, 3), (0, 5)), (1, (), ())])
def test_device_solve_of_the_reduced_system(pkg, K, lidar, no_imu):
    """k_lvi_solve alone (band of the velocity / bias unknowns eliminated four columns per step, pose block as MFMA tiles) against a dense solve and
    against the host's envelope LDL^T: window sizes from 1 to 25 keyframes (partial panels, partial tiles), keyframes without IMU states."""
    rng = np.random.default_rng(100 + K)
    Hi, S, np_, n = inertial_system(rng, K, lidar, None, no_imu)
    for lam in (1e-2, 1.0, 37.5):
        M = Hi.copy()
        M[:np_, :np_] += S
        M[np_:, np_:] += lam * np.eye(n - np_)
        rhs = rng.normal(0, 1, n)
        want = np.linalg.solve(M, rhs)
        got = pkg.capi.device_reduced_solve(Hi, S, lam, rhs)
        assert got is not None
        assert np.abs(got - want).max() <= 1e-9 * max(1.0, np.abs(want).max())
        host = pkg.capi.host_reduced_solve(Hi, S, lam, rhs)
        assert np.abs(got - host).max() <= 1e-10 * max(1.0, np.abs(host).max())
    again = pkg.capi.device_reduced_solve(Hi, S, lam, rhs)
    assert np.array_equal(again, got)   # the same bits from launch to launch


@pytest.mark.gpu
def test_device_solve_declines_and_reports(pkg):
    rng = np.random.default_rng(7)
    Hi, S, np_, n = inertial_system(rng, 8, (), (1, 6))      # an edge between keyframes five places apart: the band is wider than the rings
    with pytest.raises(pkg.capi.Tc2liError):
        pkg.capi.device_reduced_solve(Hi, S, 1.0, np.ones(n))
    Hi, S, np_, n = inertial_system(rng, 4)
    Hi[np_ + 2, :] = 0; Hi[:, np_ + 2] = 0                   # a zero pivot
    assert pkg.capi.device_reduced_solve(Hi, S, 0.0, np.ones(n)) is None
