"""Oracle SO3 primitives vs the reference's closed forms and thresholds (geometry.h:17-166)."""
import numpy as np
import pytest


def skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])


def test_exp_matches_matrix_exponential(oracle_lib):
    from scipy.linalg import expm
    rng = np.random.default_rng(0)
    for _ in range(20):
        w = rng.standard_normal(3) * rng.uniform(0.01, 3.0)
        assert np.allclose(oracle_lib.so3_exp(w), expm(skew(w)), atol=1e-13)


def test_exp_first_order_branch(oracle_lib):
    # geometry.h:137-139: below 1e-9 the result is exactly I + [v]x
    w = np.array([3e-10, -2e-10, 1e-10])
    assert np.array_equal(oracle_lib.so3_exp(w), np.eye(3) + skew(w))


def test_log_inverts_exp(oracle_lib):
    rng = np.random.default_rng(1)
    for _ in range(20):
        w = rng.standard_normal(3)
        w *= rng.uniform(0.01, 3.0) / np.linalg.norm(w)
        assert np.allclose(oracle_lib.so3_log(oracle_lib.so3_exp(w)), w, atol=1e-10)


def test_log_near_pi_uses_first_order_quirk(oracle_lib):
    # geometry.h:159-160: |sin(angle)| < 1e-9 also near pi => 0.5 * vee(M - M^T), NOT the true log
    R = oracle_lib.so3_exp(np.array([np.pi, 0, 0]))
    w = oracle_lib.so3_log(R)
    assert np.linalg.norm(w) < 1e-6  # the true log has norm pi; the reference's branch returns ~0


def test_right_jacobian_threshold_and_derivative(oracle_lib):
    # geometry.h:33-34: exactly identity below 1e-5
    assert np.array_equal(oracle_lib.so3_right_jacobian(np.array([5e-6, 0, 0])), np.eye(3))
    # exp(w + dw) ~ exp(w) exp(Jr(w) dw)
    rng = np.random.default_rng(2)
    w = rng.standard_normal(3) * 0.7
    Jr = oracle_lib.so3_right_jacobian(w)
    h = 1e-6
    for i in range(3):
        e = np.zeros(3); e[i] = h
        d = oracle_lib.so3_log(oracle_lib.so3_exp(w).T @ oracle_lib.so3_exp(w + e)) / h
        assert np.allclose(d, Jr[:, i], atol=1e-5)
