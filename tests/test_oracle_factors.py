"""Oracle factor evaluators vs the reference's own acceptance criterion: analytic Jacobians agree with
numeric ones (ceres::GradientChecker, |sum(J_a - J_n)| <= 1e-5 — residual_test.cpp:124-125,272-273;
imu_test.cpp:457-462), on the reference tests' fixture (K = diag(100,100), c = (400,400),
residual_test.cpp:26-31), plus the quirks of SURVEY.md Appendix B."""
import numpy as np
import pytest

from sadvio_amd.synthetic import T_to_12, exp_so3


def rand_pose(rng, tscale=1.0):
    T = np.eye(4)
    T[:3, :3] = exp_so3(rng.standard_normal(3))
    T[:3, 3] = tscale * rng.uniform(-1, 1, 3)
    return T


def numdiff(f, x, h=1e-6):
    x = np.asarray(x, dtype=float)
    f0 = f(x)
    J = np.zeros((f0.size, x.size))
    for i in range(x.size):
        a, b = x.copy(), x.copy()
        a[i] += h; b[i] -= h
        J[:, i] = (f(a) - f(b)) / (2 * h)
    return J


K = np.array([100.0, 100.0, 400.0, 400.0])
I12 = T_to_12(np.eye(4))


def _visible_setup(rng):
    while True:
        T = rand_pose(rng)
        p = rng.uniform(-1, 1, 3)
        pc = T[:3, :3] @ p + T[:3, 3]
        if pc[2] < 0.3:
            continue
        uv = np.array([K[0] * pc[0] / pc[2] + K[2], K[1] * pc[1] / pc[2] + K[3]])
        if 0 < uv[0] < 800 and 0 < uv[1] < 800:
            return T, p, uv


@pytest.mark.parametrize("scale", [0.0, 1e-3, 0.05])
def test_pixel_factor_jacobians(oracle_lib, scale):
    rng = np.random.default_rng(3)
    for _ in range(5):
        T, p, uv = _visible_setup(rng)
        T0 = T_to_12(T)
        meas = uv + rng.standard_normal(2)
        dpose = scale * rng.standard_normal(6); dl = scale * rng.standard_normal(3)
        r, Jp, Jl, valid = oracle_lib.factor_pixel(T0, K, I12, p, meas, 1.0, dpose, dl)
        assert valid == 1
        Jpn = numdiff(lambda x: oracle_lib.factor_pixel(T0, K, I12, p, meas, 1.0, x, dl)[0], dpose)
        Jln = numdiff(lambda x: oracle_lib.factor_pixel(T0, K, I12, p, meas, 1.0, dpose, x)[0], dl)
        assert abs((Jp - Jpn).sum()) <= 1e-5 and abs((Jl - Jln).sum()) <= 1e-5  # the reference's bar
        assert np.allclose(Jp, Jpn, rtol=1e-6, atol=1e-5) and np.allclose(Jl, Jln, rtol=1e-6, atol=1e-5)


def test_pixel_residual_is_zero_at_ground_truth(oracle_lib):
    rng = np.random.default_rng(4)
    T, p, uv = _visible_setup(rng)
    r, _, _, valid = oracle_lib.factor_pixel(T_to_12(T), K, I12, p, uv, 1.0, np.zeros(6), np.zeros(3))
    assert valid == 1 and np.abs(r).max() < 1e-10


def test_pixel_invalid_projection_zeroes_residual_keeps_jacobian(oracle_lib):
    # Quirk B.1 (…Analytic.h:63-65, Camera.cpp:128-136): behind the camera / out of [0,2cx]x[0,2cy]
    T0 = I12
    for p, uv in [(np.array([0.1, 0.1, 0.05]), np.array([400.0, 400.0])),   # z < 0.1
                  (np.array([5.0, 0.0, 1.0]), np.array([790.0, 400.0]))]:    # u = 900 > 2 cx
        r, Jp, Jl, valid = oracle_lib.factor_pixel(T0, K, I12, p, uv, 1.0, np.zeros(6), np.zeros(3))
        assert valid == 0 and np.array_equal(r, np.zeros(2))
        assert np.abs(Jp).max() > 0 and np.abs(Jl).max() > 0


def test_pixel_sigma_scales_residual_and_jacobian(oracle_lib):
    rng = np.random.default_rng(5)
    T, p, uv = _visible_setup(rng)
    a = oracle_lib.factor_pixel(T_to_12(T), K, I12, p, uv + 1.0, 1.0, np.zeros(6), np.zeros(3))
    b = oracle_lib.factor_pixel(T_to_12(T), K, I12, p, uv + 1.0, 2.0, np.zeros(6), np.zeros(3))
    assert np.allclose(a[0], 2 * b[0]) and np.allclose(a[1], 2 * b[1]) and np.allclose(a[2], 2 * b[2])


@pytest.mark.parametrize("scale", [0.0, 1e-3, 0.05])
def test_angular_factor_jacobians(oracle_lib, scale):
    rng = np.random.default_rng(6)
    for _ in range(5):
        T, p, uv = _visible_setup(rng)
        T0 = T_to_12(T)
        Tsf = T_to_12(rand_pose(rng, 0.1))
        Ts = np.eye(4); Ts[:3, :3] = Tsf[:9].reshape(3, 3); Ts[:3, 3] = Tsf[9:]
        ps = (Ts @ T @ np.append(p, 1))[:3]
        b = ps / np.linalg.norm(ps) + 0.01 * rng.standard_normal(3)
        b /= np.linalg.norm(b)
        dpose = scale * rng.standard_normal(6); dl = scale * rng.standard_normal(3)
        r, Jp, Jl = oracle_lib.factor_angular(T0, Tsf, p, b, 0.01, dpose, dl)
        Jpn = numdiff(lambda x: oracle_lib.factor_angular(T0, Tsf, p, b, 0.01, x, dl)[0], dpose)
        Jln = numdiff(lambda x: oracle_lib.factor_angular(T0, Tsf, p, b, 0.01, dpose, x)[0], dl)
        assert np.allclose(Jp, Jpn, rtol=1e-5, atol=1e-4) and np.allclose(Jl, Jln, rtol=1e-5, atol=1e-4)


def test_angular_residual_zero_and_ex_branch(oracle_lib):
    # bearing == e_x switches the tangent basis to b x e_z (…Angular….h:67-73)
    T0 = I12
    p = np.array([2.0, 0.0, 0.0])
    r, Jp, Jl = oracle_lib.factor_angular(T0, I12, p, np.array([1.0, 0, 0]), 1.0, np.zeros(6), np.zeros(3))
    assert np.abs(r).max() < 1e-14 and np.isfinite(Jp).all() and np.isfinite(Jl).all()
    Jln = numdiff(lambda x: oracle_lib.factor_angular(T0, I12, p, np.array([1.0, 0, 0]), 1.0, np.zeros(6), x)[0],
                  np.zeros(3))
    assert np.allclose(Jl, Jln, atol=1e-6)


@pytest.mark.parametrize("scale", [0.0, 0.02])
def test_pose_prior_jacobian(oracle_lib, scale):
    # residual_test.cpp:66-96 PriorResidual: PosePriordx(I, T_rand, 100 * I)
    rng = np.random.default_rng(7)
    for _ in range(5):
        Tp = T_to_12(rand_pose(rng))
        T0 = I12 if scale == 0 else T_to_12(rand_pose(rng))
        d = scale * rng.standard_normal(6)
        inf = 100 * np.ones(6)
        r, J = oracle_lib.factor_pose_prior(T0, Tp, inf, d)
        Jn = numdiff(lambda x: oracle_lib.factor_pose_prior(T0, Tp, inf, x)[0], d)
        assert abs((J - Jn).sum()) <= 1e-5 * 100  # scaled by the 100x information like the reference's sum
        assert np.allclose(J, Jn, rtol=1e-5, atol=1e-4)
    r, _ = oracle_lib.factor_pose_prior(Tp, Tp, inf, np.zeros(6))
    assert np.abs(r).max() < 1e-9
