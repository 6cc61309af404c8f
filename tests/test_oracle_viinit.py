"""Oracle of the visual-inertial initialisation (AOptimizer::VIInit, AOptimizer.cpp:448-581; IMUFactorInit,
residuals.hpp:302-410), pinned against the reference's own test of that factor (cpp/tests/imu_test.cpp:489-545):
analytic vs numeric Jacobians by the reference's criterion, and the scale recovered from ONE factor with every
block free."""
import numpy as np

from imu_helpers import arr, factor_dict
from sadvio_amd import capi
from test_oracle_imu import _free_fall_chain
from viinit_helpers import make_viinit


def _scaled_fixture(scale=0.5):  # imu_test.cpp:489-497: both poses' translations scaled
    T_i_f, ch, cur, cfg = _free_fall_chain()
    f = factor_dict(0, 1, cur, 1.0, cfg)
    Ti, Tj = arr(ch.kf.T_f_w).copy(), arr(cur.T_f_w).copy()
    Ti[9:] *= scale; Tj[9:] *= scale
    return f, Ti, Tj, arr(ch.kf.v), arr(cur.v)


def test_IMUFactorInit_jacobians_reference_criterion(oracle_lib):  # :499-530
    f, Ti, Tj, vi, vj = _scaled_fixture()
    x0 = np.zeros(15)
    r, J = oracle_lib.factor_imu_init(f, Ti, Tj, vi, vj, x0)
    h = 1e-6
    Jn = np.zeros((9, 15))
    for k in range(15):
        a = np.zeros(15); a[k] = h
        Jn[:, k] = (oracle_lib.factor_imu_init(f, Ti, Tj, vi, vj, x0 + a)[0] - oracle_lib.factor_imu_init(f, Ti, Tj, vi, vj, x0 - a)[0]) / (2 * h)
    for lo, hi in [(0, 2), (2, 5), (5, 8), (8, 11), (11, 14), (14, 15)]:   # the reference's six blocks, :525-530
        assert abs((J[:, lo:hi] - Jn[:, lo:hi]).sum()) < 1e-5 * max(1.0, np.abs(Jn[:, lo:hi]).max())
    # at a non-zero point too (the scale column is coded WITHOUT the exp(lambda) factor, residuals.hpp:398-405:
    # exact at lambda = 0, off by exp(lambda) elsewhere — kept as coded)
    x1 = np.array([0.02, -0.03, 0.1, -0.2, 0.05, 0.03, 0.02, -0.01, 0.01, 0.02, -0.01, 0.003, -0.002, 0.001, 0.3])
    r1, J1 = oracle_lib.factor_imu_init(f, Ti, Tj, vi, vj, x1)
    Jn1 = np.zeros((9, 15))
    for k in range(15):
        a = np.zeros(15); a[k] = h
        Jn1[:, k] = (oracle_lib.factor_imu_init(f, Ti, Tj, vi, vj, x1 + a)[0] - oracle_lib.factor_imu_init(f, Ti, Tj, vi, vj, x1 - a)[0]) / (2 * h)
    assert np.allclose(J1[:, :14], Jn1[:, :14], rtol=1e-5, atol=1e-5 * np.abs(Jn1).max())
    assert np.allclose(J1[:, 14] * np.exp(0.3), Jn1[:, 14], rtol=1e-5, atol=1e-5 * np.abs(Jn1).max())


def test_single_factor_recovers_the_scale(oracle_lib):  # :532-545: ASSERT_NEAR(scale, 1 / exp(lambda), 1e-2)
    scale = 0.5
    f, Ti, Tj, vi, vj = _scaled_fixture(scale)
    o = capi.viinit_options()      # that test leaves Ceres' default of 50 iterations; f_tol 1e-3 (:540)
    res = oracle_lib.viinit(np.stack([Ti, Tj]), np.stack([vi, vj]), [f], o, optim_scale=True, optim_bias=True,
                            sigma_dba=1e30, sigma_dbg=1e30)   # no bias prior in that test
    assert res["rc"] == 0
    assert abs(scale - 1.0 / np.exp(res["lambda"])) < 1e-2
    # the scale column of the Jacobian lacks the exp(lambda) factor (as coded), so LM converges slowly: all 50 iterations
    assert res["summary"].iterations == 50 and res["summary"].final_cost < 1e-6 * res["summary"].initial_cost


def test_viinit_recovers_scale_gravity_and_velocities(oracle_lib):
    """The reference's window-level acceptance (imu_test.cpp:858-880: poses back on the ground truth to 0.02 after
    VIInit with optim_scale) on a synthetic 10-key-frame trajectory."""
    pb = make_viinit(n_kf=10, scale=0.5, tilt=(0.05, -0.08))
    res = oracle_lib.viinit(pb["T_f_w"], pb["vel"], pb["factors"], capi.viinit_options(), optim_scale=True)
    assert res["rc"] == 0 and res["summary"].iterations <= 50
    assert abs(res["scale"] - pb["truth"]["scale"]) < 2e-3 * pb["truth"]["scale"]
    assert np.abs(res["R_w_i"] - pb["truth"]["R_w_i"]).max() < 2e-3
    assert np.abs(pb["vel"] + res["dv"] - pb["truth"]["vel"]).max() < 5e-3
    assert res["summary"].final_cost < 1e-3 * res["summary"].initial_cost
    # frames no factor touches keep a zero velocity delta; scale constant => lambda stays 0
    res0 = oracle_lib.viinit(pb["T_f_w"], pb["vel"], pb["factors"][:3], capi.viinit_options(), optim_scale=False)
    assert res0["lambda"] == 0.0 and res0["scale"] == 1.0
    touched = sorted({f["kf_i"] for f in pb["factors"][:3]} | {f["kf_j"] for f in pb["factors"][:3]})
    untouched = [k for k in range(len(pb["vel"])) if k not in touched]
    assert np.abs(res0["dv"][untouched]).max() == 0.0 and np.abs(res0["dv"][touched]).max() > 0.0


def test_viinit_empty_problem(oracle_lib):
    pb = make_viinit(n_kf=3)
    res = oracle_lib.viinit(pb["T_f_w"], pb["vel"], [], capi.viinit_options(), optim_scale=True)
    assert res["rc"] == 0 and res["summary"].iterations == 0 and res["scale"] == 1.0 and np.abs(res["dv"]).max() == 0.0
