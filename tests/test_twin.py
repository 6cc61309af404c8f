"""The C oracle against its independent twin (oracle/twin.py: NumPy float64 / long double / 50-digit mpmath, written from
the reference's source lines and Ceres 2.2.0's published trust-region algorithm, un-reduced normal equations):
factor arithmetic at 50 digits, one full LM iteration at 50 digits, complete solves iterate by iterate. This is the pin of
the headline localMapBA path, for which the reference holds no golden vector."""
import numpy as np
import pytest

from oracle import twin
from sadvio_amd import capi, synthetic

mp = pytest.importorskip("mpmath")


def _rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


@pytest.mark.parametrize("scale", [1.0, 1e-3, 1e-7, 0.0])   # incl. the Jr = I (< 1e-5) and first-order exp / log (< 1e-9) branches
def test_factors_match_50_digit_evaluation(oracle_lib, scale):
    rng = np.random.default_rng(11)
    w = synthetic.make_window(n_kf=4, n_lmk=12, obs_per_lmk=4, seed=9)
    B = twin.Backend("mp", 50)
    worst = 0.0
    for o in range(0, w.n_obs, 3):
        l = int(np.searchsorted(w.lmk_obs_ptr, o, side="right") - 1)
        k, c = int(w.obs_kf[o]), int(w.obs_cam[o])
        dp = scale * np.concatenate([0.05 * rng.standard_normal(3), 0.1 * rng.standard_normal(3)])
        dl = scale * 0.05 * rng.standard_normal(3)
        r, Jp, Jl, v = oracle_lib.factor_pixel(w.kf_T_f_w[k], w.cam_K[c], w.cam_T_s_f[c], w.lmk_p[l], w.obs_meas[o], 1.0, dp, dl)
        rm, Jpm, Jlm, vm = twin.pixel_factor(B, w.kf_T_f_w[k], w.cam_K[c], w.cam_T_s_f[c], w.lmk_p[l], w.obs_meas[o], 1.0, dp, dl)
        assert bool(v) == bool(vm)
        worst = max(worst, _rel(Jp, B.f(Jpm)), _rel(Jl, B.f(Jlm)), float(np.abs(r - B.f(rm)).max()) / 1e3)
        b = np.array([(w.obs_meas[o][0] - w.cam_K[c][2]) / w.cam_K[c][0], (w.obs_meas[o][1] - w.cam_K[c][3]) / w.cam_K[c][1], 1.0])
        b /= np.linalg.norm(b)
        r, Jp, Jl = oracle_lib.factor_angular(w.kf_T_f_w[k], w.cam_T_s_f[c], w.lmk_p[l], b, 0.003, dp, dl)
        rm, Jpm, Jlm = twin.angular_factor(B, w.kf_T_f_w[k], w.cam_T_s_f[c], w.lmk_p[l], b, 0.003, dp, dl)
        worst = max(worst, _rel(Jp, B.f(Jpm)), _rel(Jl, B.f(Jlm)), float(np.abs(r - B.f(rm)).max()) / 1e3)
        Tp = w.kf_T_f_w[(k + 1) % w.n_kf]
        inf = np.array([100.0, 50.0, 20.0, 10.0, 5.0, 1.0])
        r, J = oracle_lib.factor_pose_prior(w.kf_T_f_w[k], Tp, inf, dp)
        rm, Jm = twin.pose_prior_factor(B, w.kf_T_f_w[k], Tp, inf, dp)
        worst = max(worst, _rel(J, B.f(Jm)), _rel(r, B.f(rm)))
    assert worst < 5e-13, worst


def test_invalid_projection_branch_matches(oracle_lib):
    """Camera.cpp:127-137: behind the camera / outside [0, 2c] -> residual 0, Jacobians kept."""
    w = synthetic.make_window(n_kf=3, n_lmk=6, obs_per_lmk=3, seed=2)
    B = twin.Backend("f64")
    k, c = int(w.obs_kf[0]), int(w.obs_cam[0])
    R, t = w.kf_T_f_w[k][:9].reshape(3, 3), w.kf_T_f_w[k][9:]
    Rs, ts = w.cam_T_s_f[c][:9].reshape(3, 3), w.cam_T_s_f[c][9:]
    for p_cam in (np.array([0.1, 0.2, -5.0]), np.array([0.1, 0.2, 0.05]), np.array([9.0, 0.0, 4.0]), np.array([0.0, -6.0, 4.0])):
        p = R.T @ (Rs.T @ (p_cam - ts) - t)      # behind the camera / closer than 0.1 / outside the image on either axis
        r, Jp, Jl, v = oracle_lib.factor_pixel(w.kf_T_f_w[k], w.cam_K[c], w.cam_T_s_f[c], p, w.obs_meas[0], 1.0, np.zeros(6), np.zeros(3))
        rt, Jpt, Jlt, vt = twin.pixel_factor(B, w.kf_T_f_w[k], w.cam_K[c], w.cam_T_s_f[c], p, w.obs_meas[0], 1.0, np.zeros(6), np.zeros(3))
        assert not v and not vt and np.all(r == 0) and np.all(rt == 0)
        assert _rel(Jp, Jpt) < 1e-12 and _rel(Jl, Jlt) < 1e-12 and np.abs(Jp).max() > 0


@pytest.mark.parametrize("factor", [capi.FACTOR_PIXEL, capi.FACTOR_ANGULAR])
def test_first_lm_iteration_matches_50_digit_unreduced_solve(oracle_lib, factor):
    """The oracle's first step (Schur complement + Cholesky in float64) against (J^T J + D^2) y = J^T r solved at 50 digits
    on the un-reduced system, with Ceres' Jacobi scaling and LM diagonal."""
    w = synthetic.make_window(n_kf=3, n_lmk=14, obs_per_lmk=4, seed=5, factor=factor)
    opts = capi.reference_options()
    ref = twin.first_iteration(w, opts, kind="mp")
    dp, dl, H, g = oracle_lib.first_step(w, opts)
    assert np.abs(dp - ref["pose"]).max() < 1e-10 * max(1.0, np.abs(ref["pose"]).max())
    assert np.abs(dl - ref["lmk"]).max() < 1e-10 * max(1.0, np.abs(ref["lmk"]).max())
    sol = oracle_lib.solve(w, capi.gn_options(1))
    assert np.isclose(sol["log"][1][7], ref["log"][1][7], rtol=1e-11)   # model cost change
    assert np.isclose(sol["log"][1][0], ref["log"][1][0], rtol=1e-11)   # cost after the step
    assert np.isclose(sol["log"][1][4], ref["log"][1][4], rtol=1e-9)    # step quality rho


def _variants():
    w = synthetic.make_window(n_kf=5, n_lmk=50, obs_per_lmk=5, seed=3)
    yield "pixel", w, capi.reference_options()
    w = synthetic.make_window(n_kf=5, n_lmk=50, obs_per_lmk=4, seed=4, factor=capi.FACTOR_ANGULAR)
    yield "angular", w, capi.reference_options()
    w = synthetic.make_window(n_kf=4, n_lmk=40, obs_per_lmk=5, seed=6, fixed=0)   # no constant frame: the pose prior holds the gauge
    w.lmk_const = (np.arange(w.n_lmk) % 7 == 0).astype(np.uint8)
    yield "free-gauge + constant landmarks", w, capi.reference_options()
    w = synthetic.make_window(n_kf=4, n_lmk=40, obs_per_lmk=5, seed=7, fixed=2, pixel_noise=3.0)
    o = capi.reference_options(); o.huber_a = 1.345 ** 0.5
    yield "huber, two constant frames", w, o
    w = synthetic.make_window(n_kf=4, n_lmk=40, obs_per_lmk=4, seed=8, lmk_perturb=0.3, rot_perturb_deg=2.0)   # rejected steps
    o = capi.reference_options(); o.initial_trust_region_radius = 1e-2; o.function_tolerance = 1e-6
    yield "small radius (growing trust region), tight tolerance", w, o
    rng = np.random.default_rng(5)
    w = synthetic.make_window(n_kf=4, n_lmk=40, obs_per_lmk=5, seed=9)
    li = np.array([3, 8, 20, 31], dtype=np.int32)
    n = 3 * len(li)
    w.dense_prior = {"J": rng.standard_normal((n - 2, n)), "r0": 0.3 * rng.standard_normal(n - 2), "kf_keep": -1, "kf_col": 0,
                     "lmk_index": li, "lmk_col": np.arange(0, n, 3, dtype=np.int32)}
    yield "VO dense prior", w, capi.reference_options()


@pytest.mark.parametrize("name,w,opts", list(_variants()), ids=[v[0] for v in _variants()])
def test_full_solve_matches_twin_iterate_by_iterate(oracle_lib, name, w, opts):
    ref = twin.lm_solve(w, opts, kind="f64")          # dense LAPACK solve of the un-reduced system
    got = oracle_lib.solve(w, opts, dense_prior=w.dense_prior)
    s = got["summary"]
    assert (s.iterations, s.termination, s.num_successful_steps, s.num_unsuccessful_steps) == \
        (ref["iterations"], ref["termination"], ref["n_success"], ref["n_unsuccess"])
    assert np.isclose(s.initial_cost, ref["initial_cost"], rtol=1e-12) and np.isclose(s.final_cost, ref["final_cost"], rtol=1e-10)
    assert np.isclose(s.fixed_cost, ref["fixed_cost"], rtol=1e-12, atol=1e-12)
    L, T = got["log"], ref["log"]
    assert L.shape == T.shape
    n = len(L) - (1 if s.termination in (1, 2) else 0)   # the terminating attempt is logged differently (candidate not accepted)
    assert np.allclose(L[:n, 0], T[:n, 0], rtol=1e-10)               # cost after each iteration  (1e-9)
    assert np.allclose(L[:, 1], T[:, 1], rtol=1e-7, atol=1e-9 * s.initial_cost)   # cost change
    assert np.allclose(L[:n, 2], T[:n, 2], rtol=1e-7)                # trust-region radius
    assert np.allclose(L[:, 3], T[:, 3], rtol=1e-8)                  # step norm
    assert np.allclose(L[:n, 6], T[:n, 6], rtol=1e-8)                # max |gradient|
    assert np.allclose(L[:, 7], T[:, 7], rtol=1e-8)                  # model cost change
    assert np.abs(got["pose"] - ref["pose"]).max() < 1e-9
    assert np.abs(got["lmk"] - ref["lmk"]).max() < 1e-8 * max(1.0, np.abs(ref["lmk"]).max())
