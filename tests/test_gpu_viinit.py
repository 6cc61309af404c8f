"""GPU parity of the visual-inertial initialisation (AOptimizer::VIInit, AOptimizer.cpp:448-581):
the single-workgroup LM solve of k_viinit against the oracle (oracle/viinit.c) through the C ABI."""
import numpy as np
import pytest

from sadvio_amd import capi
from test_oracle_viinit import _scaled_fixture
from viinit_helpers import make_viinit

pytestmark = pytest.mark.gpu


def compare(be, oracle_lib, T, v, factors, opts=None, tol=1e-8, **kw):
    opts = opts or capi.viinit_options()
    got = be.vi_init(T, v, factors, opts, **kw)
    ref = oracle_lib.viinit(T, v, factors, opts, **kw)
    gs, rs = got["summary"], ref["summary"]
    assert got["rc"] == ref["rc"]
    assert np.isclose(gs.initial_cost, rs.initial_cost, rtol=1e-10, atol=1e-14)
    assert (gs.iterations, gs.termination, gs.num_successful_steps, gs.num_unsuccessful_steps) == \
           (rs.iterations, rs.termination, rs.num_successful_steps, rs.num_unsuccessful_steps)
    assert np.isclose(gs.final_cost, rs.final_cost, rtol=1e-6, atol=1e-12 * max(rs.initial_cost, 1.0))
    for k in ("r_wi", "dba", "dbg", "dv", "R_w_i"):
        assert np.abs(got[k] - ref[k]).max() <= tol, k
    assert abs(got["lambda"] - ref["lambda"]) <= tol and abs(got["scale"] - ref["scale"]) <= 10 * tol
    return got, ref


def test_reference_factor_test_scale_recovery(backend_cls, oracle_lib):
    """imu_test.cpp:489-545 on the device: one IMUFactorInit, every block free, scale 0.5 recovered to 1e-2."""
    f, Ti, Tj, vi, vj = _scaled_fixture(0.5)
    be = backend_cls(device=0)
    try:
        got, _ = compare(be, oracle_lib, np.stack([Ti, Tj]), np.stack([vi, vj]), [f], optim_scale=True, optim_bias=True,
                         sigma_dba=1e30, sigma_dbg=1e30)
    finally:
        be.close()
    assert abs(0.5 - 1.0 / got["scale"]) < 1e-2


@pytest.mark.parametrize("n_kf,optim_scale,noise", [(10, True, 0.0), (10, False, 0.02), (25, True, 0.02), (48, True, 0.01)])
def test_viinit_windows(backend_cls, oracle_lib, n_kf, optim_scale, noise):
    pb = make_viinit(n_kf=n_kf, scale=0.5 if optim_scale else 1.0, tilt=(0.05, -0.08), vel_noise=noise, seed=n_kf)
    be = backend_cls(device=0)
    try:
        got, _ = compare(be, oracle_lib, pb["T_f_w"], pb["vel"], pb["factors"], optim_scale=optim_scale)
    finally:
        be.close()
    assert abs(got["scale"] - pb["truth"]["scale"]) < 5e-3 * pb["truth"]["scale"]
    assert np.abs(got["R_w_i"] - pb["truth"]["R_w_i"]).max() < 5e-3


def test_viinit_free_biases_with_priors_and_untouched_frames(backend_cls, oracle_lib):
    pb = make_viinit(n_kf=12, scale=0.7, tilt=(-0.03, 0.06), vel_noise=0.01, seed=5)
    be = backend_cls(device=0)
    try:
        got, _ = compare(be, oracle_lib, pb["T_f_w"], pb["vel"], pb["factors"], optim_scale=True, optim_bias=True,
                         sigma_dba=0.05, sigma_dbg=0.01)
        # only the 4 newest factors: 7 frames carry no factor and keep a zero velocity delta
        got2, _ = compare(be, oracle_lib, pb["T_f_w"], pb["vel"], pb["factors"][:4], optim_scale=True)
        touched = sorted({f["kf_i"] for f in pb["factors"][:4]} | {f["kf_j"] for f in pb["factors"][:4]})
        untouched = [k for k in range(12) if k not in touched]
        assert np.abs(got2["dv"][untouched]).max() == 0.0
        # empty program
        got3 = be.vi_init(pb["T_f_w"], pb["vel"], [], capi.viinit_options(), optim_scale=True)
        assert got3["rc"] == 0 and got3["summary"].iterations == 0 and got3["scale"] == 1.0
    finally:
        be.close()


def test_viinit_rejects_bad_input(backend_cls):
    pb = make_viinit(n_kf=4)
    be = backend_cls(device=0)
    try:
        bad = [dict(pb["factors"][0], kf_j=9)]
        with pytest.raises(RuntimeError):
            be.vi_init(pb["T_f_w"], pb["vel"], bad)
        with pytest.raises(RuntimeError):
            be.vi_init(np.tile(pb["T_f_w"][:1], (49, 1)), np.zeros((49, 3)), [])
    finally:
        be.close()
