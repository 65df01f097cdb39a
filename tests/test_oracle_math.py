"""Oracle math primitives vs numpy/scipy (SURVEY §4 test plan item 1)."""
import numpy as np
import pytest
from scipy.linalg import expm

import oracle as O
import np_reference as NP


def test_se3_exp_matches_matrix_exponential():
    rng = np.random.default_rng(0)
    for scale in (1e-12, 1e-6, 1e-2, 1.0, 3.0):
        d = rng.normal(size=6) * scale
        X = np.zeros((4, 4))
        X[:3, :3] = NP.skew(d[:3])
        X[:3, 3] = d[3:]
        assert np.allclose(O.se3_exp(d), expm(X), atol=1e-12)
        assert np.allclose(O.se3_exp(d), NP.se3_exp(d), atol=1e-12)


@pytest.mark.parametrize("svd", [False, True])
def test_solve6(svd):
    rng = np.random.default_rng(1)
    for _ in range(20):
        A = rng.normal(size=(6, 6))
        A = A @ A.T + 1e-3 * np.eye(6)
        A *= 10.0 ** rng.uniform(-3, 6)
        b = rng.normal(size=6)
        x = O.solve6(A, b, svd=svd)
        assert np.allclose(x, np.linalg.solve(A, b), rtol=1e-8, atol=1e-12)
    # symmetric negative definite (the NDT Hessian near an optimum)
    A = -(A @ A.T + np.eye(6))
    assert np.allclose(O.solve6(A, b, svd=True), np.linalg.solve(A, b), rtol=1e-8)


def test_solve_svd_rank_deficient_is_min_norm():
    rng = np.random.default_rng(2)
    B = rng.normal(size=(6, 4))
    A = B @ B.T  # rank 4
    b = A @ rng.normal(size=6)
    x = O.solve6(A, b, svd=True)
    assert np.allclose(x, np.linalg.pinv(A) @ b, atol=1e-9)


def test_eig_sym3():
    rng = np.random.default_rng(3)
    for _ in range(50):
        B = rng.normal(size=(3, 3))
        A = B @ B.T * 10.0 ** rng.uniform(-4, 2)
        w, V = O.eig_sym3(A)
        wr = np.linalg.eigvalsh(A)
        assert np.allclose(w, wr, rtol=1e-10, atol=1e-14)
        assert np.allclose(V @ np.diag(w) @ V.T, A, atol=1e-12 * max(1, np.abs(A).max()))
    w, V = O.eig_sym3(np.diag([3.0, 1.0, 2.0]))
    assert np.allclose(w, [1, 2, 3])


def test_euler_xyz_reproduces_rotation_and_eigen_range():
    rng = np.random.default_rng(4)
    for _ in range(100):
        p = np.concatenate([np.zeros(3), rng.uniform(-0.6, 0.6, size=3)])
        R = O.ndt_pose(p)[:3, :3]
        e = O.euler_xyz(R.astype(np.float32))
        assert 0.0 <= e[0] <= np.pi + 1e-6          # Eigen::eulerAngles first-angle range
        R2 = O.ndt_pose(np.concatenate([np.zeros(3), e]))[:3, :3]
        assert np.allclose(R, R2, atol=2e-6)
        assert np.allclose(O.ndt_pose(p), NP.ndt_pose(p), atol=1e-15)
