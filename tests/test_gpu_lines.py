"""SURVEY.md §8f rank 3 on the GPU: linexd line landmarks (ReprojectionErrCeres_linexd_dx / AngularErrCeres_linexd_dx) in the
window solve, against the oracle (tests/test_oracle_lines.py pins that side on the reference's formulas): same LM trace,
key-frame deltas within 1e-6, line deltas within 1e-6."""
import numpy as np
import pytest

from sadvio_amd import capi
from sadvio_amd.synthetic import make_window
from golden_util import assert_trace_matches
from line_helpers import add_lines

pytestmark = pytest.mark.gpu
TOL = 1e-6


def run_both(backend_cls, oracle_lib, w, opts, use_graph=False):
    be = backend_cls(device=0, use_graph=use_graph)
    try:
        be.set_windows([w])
        s = be.solve(opts)[0]
        d = be.get_deltas(0)
        dl = be.get_line_deltas(0, w.lines["T_w_l"].shape[0])
        trace = be.get_trace(0)
    finally:
        be.close()
    ref = oracle_lib.solve(w, opts)
    rs = ref["summary"]
    assert (s.iterations, s.termination, s.num_successful_steps) == (rs.iterations, rs.termination, rs.num_successful_steps)
    assert np.isclose(s.initial_cost, rs.initial_cost, rtol=1e-11)
    assert np.isclose(s.final_cost, rs.final_cost, rtol=1e-8)
    assert_trace_matches(trace, ref["log"], rs.termination)
    assert np.abs(d["pose"] - ref["pose"]).max() <= TOL
    assert np.abs(d["lmk"] - ref["lmk"]).max() <= TOL
    assert np.abs(dl - ref["line"]).max() <= TOL
    return s, dl, ref


@pytest.mark.parametrize("factor", [capi.FACTOR_PIXEL, capi.FACTOR_ANGULAR])
@pytest.mark.parametrize("use_graph", [False, True])
def test_window_with_lines_matches_oracle(backend_cls, oracle_lib, factor, use_graph):
    w = add_lines(make_window(n_kf=8, n_lmk=600, obs_per_lmk=4, seed=21, factor=factor), n_line=7, obs_per_line=5, n_const=2)
    opts = capi.reference_options()
    if factor == capi.FACTOR_ANGULAR:
        opts.max_num_iterations = 2     # see test_lines_with_huber_loss: later iterations linearise where log_so3 is ill-conditioned
    s, dl, ref = run_both(backend_cls, oracle_lib, w, opts, use_graph)
    assert np.all(dl[:2] == 0.0) and np.abs(dl[2:]).max() > 0.0
    assert s.final_cost < s.initial_cost


@pytest.mark.parametrize("factor", [capi.FACTOR_PIXEL, capi.FACTOR_ANGULAR])
def test_lines_with_huber_loss(backend_cls, oracle_lib, factor):
    """The line blocks carry the caller's loss function (…Analytic.cpp:303-306). Angular case: two iterations only. The
    rotation of a line about its own axis is a gauge of the angular residual, its coded Jacobian column is exactly zero at
    w = 0 and not afterwards (it goes through so3_rightJacobian(log_so3(exp(w)))), so from the second step on that component
    runs away by many radians per step; where |w| mod 2 pi passes pi, log_so3 amplifies rounding differences between any two
    implementations without bound. Two iterations linearise at |w| < pi only."""
    w = add_lines(make_window(n_kf=6, n_lmk=300, obs_per_lmk=4, seed=5, factor=factor), n_line=5, obs_per_line=4, pert_t=0.05, pert_rot=0.03)
    opts = capi.reference_options()
    opts.huber_a = 1.0 if factor == capi.FACTOR_PIXEL else 0.02
    if factor == capi.FACTOR_ANGULAR:
        opts.max_num_iterations = 2
    s, dl, ref = run_both(backend_cls, oracle_lib, w, opts)
    assert s.num_successful_steps >= 2


def test_lines_only_move_when_key_frames_are_fixed(backend_cls, oracle_lib):
    """landmarkOptimization-style window: every key-frame constant, the reduced system is the lines' 6 x 6 blocks."""
    w = add_lines(make_window(n_kf=5, n_lmk=200, obs_per_lmk=4, seed=2, factor=capi.FACTOR_ANGULAR), n_line=4, obs_per_line=5)
    w.kf_const = np.ones(w.n_kf, dtype=np.uint8)
    w.pose_priors = []
    opts = capi.reference_options()
    opts.max_num_iterations = 2         # angular line factor: see test_lines_with_huber_loss
    s, dl, ref = run_both(backend_cls, oracle_lib, w, opts)
    assert np.abs(dl).max() > 0.0


def test_lines_in_a_window_solved_out_of_lds(backend_cls, oracle_lib):
    """36 key-frames: N_p = 6 * 35 + 6 * 6 = 246 > 174, the lines make the banded system dense (wide-panel solver)."""
    w = add_lines(make_window(n_kf=36, n_lmk=1500, obs_per_lmk=4, seed=9, factor=capi.FACTOR_PIXEL, band=4, length=30.0), n_line=6, obs_per_line=6)
    run_both(backend_cls, oracle_lib, w, capi.reference_options())


def test_two_windows_one_with_lines(backend_cls, oracle_lib):
    w0 = make_window(n_kf=6, n_lmk=300, obs_per_lmk=4, seed=31)
    w1 = add_lines(make_window(n_kf=7, n_lmk=300, obs_per_lmk=4, seed=32), n_line=3, obs_per_line=4)
    be = backend_cls(device=0)
    try:
        be.set_windows([w0, w1])
        ss = be.solve(capi.reference_options())
        d0, d1 = be.get_deltas(0), be.get_deltas(1)
        dl = be.get_line_deltas(1, 3)
    finally:
        be.close()
    for w, s, d in ((w0, ss[0], d0), (w1, ss[1], d1)):
        ref = oracle_lib.solve(w, capi.reference_options())
        assert s.iterations == ref["summary"].iterations
        assert np.abs(d["pose"] - ref["pose"]).max() <= TOL
        if w is w1:
            assert np.abs(dl - ref["line"]).max() <= TOL


def test_set_lines_validation(backend_cls):
    w = add_lines(make_window(n_kf=4, n_lmk=100, obs_per_lmk=4, seed=1), n_line=2, obs_per_line=3)
    be = backend_cls(device=0)
    try:
        bad = dict(w.lines); bad["obs_kf"] = w.lines["obs_kf"].copy(); bad["obs_kf"][0] = 99
        lines, w.lines = w.lines, None
        be.set_windows([w])
        with pytest.raises(RuntimeError):
            be.set_lines(0, bad)
        be.set_lines(0, lines)
        be.set_lines(0, None)       # cleared again: plain window
        s = be.solve(capi.reference_options())[0]
        assert s.final_cost < s.initial_cost
    finally:
        be.close()
