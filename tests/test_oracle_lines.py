"""CPU tests of the linexd factors of the oracle: ReprojectionErrCeres_linexd_dx
(BundleAdjustmentCERESAnalytic.h:104-195) and AngularErrCeres_linexd_dx (AngularAdjustmentCERESAnalytic.h:368-469)."""
import numpy as np
import pytest

from oracle import oracle
from sadvio_amd import capi
from sadvio_amd.synthetic import make_window
from line_helpers import add_lines


def _window(factor, **kw):
    w = make_window(n_kf=5, n_lmk=60, obs_per_lmk=4, seed=3, factor=factor)
    return add_lines(w, **kw)


def _opts():
    o = capi.reference_options()
    o.max_num_iterations = 10
    return o


def _numeric(w, l, o, which, eps=1e-6):
    """Central-difference Jacobian of line observation o w.r.t. the key-frame (which = 0) or line (which = 1) 6-vector."""
    kf = int(w.lines["obs_kf"][o])
    n_line = w.lines["T_w_l"].shape[0]
    cols = []
    for c in range(6):
        rs = []
        for sgn in (+1, -1):
            xp = np.zeros((w.n_kf, 6)); xl = np.zeros((n_line, 6))
            (xp[kf] if which == 0 else xl[l])[c] = sgn * eps
            rs.append(oracle.line_factor(w, l, o, xp, xl)[0])
        cols.append((rs[0] - rs[1]) / (2 * eps))
    return np.array(cols).T


@pytest.mark.parametrize("factor", [capi.FACTOR_PIXEL, capi.FACTOR_ANGULAR])
def test_zero_residual_at_truth(factor):
    w = _window(factor, noise_px=0.0, pert_rot=0.0, pert_t=0.0)
    w.kf_T_f_w = w.truth["T_f_w"].copy()
    for l in range(w.lines["T_w_l"].shape[0]):
        for o in range(w.lines["obs_ptr"][l], w.lines["obs_ptr"][l + 1]):
            r, _ = oracle.line_factor(w, l, o)
            assert np.abs(r).max() < 1e-9


def test_pixel_keyframe_block_matches_numeric_and_line_block_is_as_coded():
    w = _window(capi.FACTOR_PIXEL)
    for l in range(3):
        for o in range(w.lines["obs_ptr"][l], w.lines["obs_ptr"][l + 1]):
            r, J = oracle.line_factor(w, l, o)
            assert r.shape == (4,)
            Jn = _numeric(w, l, o, 0)
            assert np.abs(J[:, :6] - Jn).max() <= 1e-5 * max(1.0, np.abs(Jn).max())
            # the parameter is read as a translation in the line frame (…Analytic.h:121), so the residual does not see
            # components 3..5, while the coded Jacobian has [-R [pt]x | I] (…Analytic.h:156-160): translation columns = the
            # projection Jacobian w.r.t. the WORLD point
            Jl = _numeric(w, l, o, 1)
            assert np.abs(Jl[:, 3:]).max() == 0.0
            R = w.lines["T_w_l"][l][:9].reshape(3, 3)
            assert np.abs(J[:, 9:12] @ R - Jl[:, :3]).max() <= 1e-5 * max(1.0, np.abs(Jl).max())


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def _angular_line_as_coded(T_sf, T_fw, T_wl, b0, b1):
    """Matrix-form NumPy evaluation, at zero deltas, of the formulas AngularErrCeres_linexd_dx writes
    (AngularAdjustmentCERESAnalytic.h:378-459) with the helper definitions of utilities/geometry.h:327-343 as coded:
    J_norm(X) = X^T / |X|, J_normalization(X) = (I - X X^T) / |X| (X not normalised), J_AcrossX(A) = -[A]x."""
    Jnz = lambda X: (np.eye(3) - np.outer(X, X)) / np.linalg.norm(X)
    T_sl = T_sf @ T_fw @ T_wl
    R_sw = (T_sf @ T_fw)[:3, :3]
    t = T_sl[:3, 3]
    n_obs = np.cross(b0, b1); n_obs /= np.linalg.norm(n_obs)
    b = t / np.linalg.norm(t)
    d = T_sl[:3, 0] / np.linalg.norm(T_sl[:3, 0])
    n_l = np.cross(b, d); n_lh = n_l / np.linalg.norm(n_l)
    cx = np.cross(n_obs, n_lh)
    r = np.array([np.linalg.norm(cx), n_obs @ b])
    J_e0 = (cx / np.linalg.norm(cx))[None, :] @ (-_skew(n_obs)) @ Jnz(n_l)
    J_e1 = n_obs[None, :] @ Jnz(t)
    Jt_dT = np.hstack([-R_sw @ _skew(T_wl[:3, 3]), R_sw])
    JR_dT = np.hstack([-R_sw @ _skew(T_wl[:3, 0]), np.zeros((3, 3))])
    Jt_dL = np.hstack([np.zeros((3, 3)), T_sl[:3, :3]])
    JR_dL = np.hstack([-T_sl[:3, :3] @ _skew(np.array([1.0, 0, 0])), np.zeros((3, 3))])
    out = []
    for Jt, JR in ((Jt_dT, JR_dT), (Jt_dL, JR_dL)):
        out.append(np.vstack([J_e0 @ (_skew(T_sl[:3, 0]).T @ (Jnz(t) @ Jt) + _skew(n_lh) @ JR), J_e1 @ Jt]))
    return r, np.hstack(out)


def test_angular_factor_matches_matrix_form_of_the_reference_formulas():
    from sadvio_amd.synthetic import T12_to_4
    w = _window(capi.FACTOR_ANGULAR)
    for l in range(w.lines["T_w_l"].shape[0]):
        for o in range(w.lines["obs_ptr"][l], w.lines["obs_ptr"][l + 1]):
            r, J = oracle.line_factor(w, l, o)
            assert r.shape == (2,)
            kf, cam = w.lines["obs_kf"][o], w.lines["obs_cam"][o]
            m = w.lines["obs_meas"][o]
            r2, J2 = _angular_line_as_coded(T12_to_4(w.cam_T_s_f[cam]), T12_to_4(w.kf_T_f_w[kf]), T12_to_4(w.lines["T_w_l"][l]), m[:3], m[3:])
            assert np.abs(r - r2).max() < 1e-13
            assert np.abs(J - J2).max() < 1e-12 * max(1.0, np.abs(J2).max())


def test_angular_distance_row_is_exact_at_unit_range():
    """Row 1 (n_obs . t / |t|) uses J_normalization(t), which is the true derivative only for |t| = 1: with the line centre at
    unit range from the sensor the coded Jacobian equals the numerical one (both blocks)."""
    from sadvio_amd.synthetic import T12_to_4, T_to_12
    w = _window(capi.FACTOR_ANGULAR, n_line=2)
    for l in range(2):
        o = int(w.lines["obs_ptr"][l])
        kf, cam = w.lines["obs_kf"][o], w.lines["obs_cam"][o]
        T_sw = T12_to_4(w.cam_T_s_f[cam]) @ T12_to_4(w.kf_T_f_w[kf])
        T_wl = T12_to_4(w.lines["T_w_l"][l])
        t_s = (T_sw @ T_wl)[:3, 3]
        T_wl[:3, 3] = (np.linalg.inv(T_sw) @ np.array([*(t_s / np.linalg.norm(t_s)), 1.0]))[:3]
        w.lines["T_w_l"][l] = T_to_12(T_wl)
        _, J = oracle.line_factor(w, l, o)
        for which in (0, 1):
            Jn = _numeric(w, l, o, which)
            assert np.abs(J[1, 6 * which:6 * which + 6] - Jn[1]).max() < 1e-6


@pytest.mark.parametrize("factor", [capi.FACTOR_PIXEL, capi.FACTOR_ANGULAR])
def test_solve_with_lines_reduces_cost_and_moves_lines(factor):
    w = _window(factor, n_line=5, obs_per_line=4, n_const=1)
    out = oracle.solve(w, _opts())
    s = out["summary"]
    assert s.final_cost < s.initial_cost
    assert out["line"].shape == (5, 6)
    assert np.all(out["line"][0] == 0.0)             # constant line
    assert np.abs(out["line"][1:]).max() > 0.0
    # same window without lines gives a different (smaller) cost: the line factors are in the sum
    w2 = _window(factor, n_line=5, obs_per_line=4)
    w2.lines = None
    s2 = oracle.solve(w2, _opts())["summary"]
    assert s2.initial_cost < s.initial_cost
