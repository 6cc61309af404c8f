"""GPU parity of the dense marginalisation prior (MarginalizationFactor, marginalization.hpp:88-218): r = r0 + J dx over
the kept key-frame's 15 states and the kept landmarks, which stay in the reduced system instead of being eliminated.
Checked against the CPU oracle on the same inputs, on the LDS-resident path (N_p <= 174) and the HBM-resident one."""
import numpy as np
import pytest

from golden_util import assert_trace_matches, cached_oracle_solve, lmk_err
from sadvio_amd import capi, synthetic
from vio_helpers import make_vio_window

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-6
LMK_TOL = 1e-6   # = the pose bar; relative for landmarks that move by more than a metre (golden_util.lmk_err)


def random_prior(w, n_keep, kf_keep, rng, scale=3.0, rank_deficit=2):
    """A prior with the shape the reference produces: n = (15) + 3 n_keep columns, n_full <= n rows (rank-revealing
    decomposition, marginalization.cpp:318-342), one skipped landmark (lmk_col = -1, marginalization.hpp:138)."""
    cand = rng.permutation(w.n_lmk)[: n_keep + 1]
    lmk_index = np.sort(cand).astype(np.int32)
    lmk_col = np.full(len(lmk_index), -1, dtype=np.int32)
    col = 0
    kf_col = 0
    if kf_keep >= 0:
        col = 15
    for i in range(len(lmk_index)):
        if i == 1:
            continue  # skipped
        lmk_col[i] = col
        col += 3
    n = col
    nf = n - rank_deficit
    J = scale * rng.standard_normal((nf, n)) / np.sqrt(n)
    r0 = 0.5 * rng.standard_normal(nf)
    return {"J": J, "r0": r0, "kf_keep": kf_keep, "kf_col": kf_col, "lmk_index": lmk_index, "lmk_col": lmk_col}


def compare(backend_cls, oracle_lib, w, opts, vio=False, golden=None):
    be = backend_cls(device=0)
    try:
        be.set_windows([w])
        s = be.solve(opts)[0]
        d = be.get_deltas(0)
        trace = be.get_trace(0)
    finally:
        be.close()
    ref = (cached_oracle_solve(golden, oracle_lib, w, opts, dense_prior=w.dense_prior) if golden
           else oracle_lib.solve(w, opts, dense_prior=w.dense_prior))
    rs = ref["summary"]
    assert np.isclose(s.initial_cost, rs.initial_cost, rtol=1e-10)
    assert np.isclose(s.final_cost, rs.final_cost, rtol=1e-9)
    assert (s.iterations, s.termination, s.num_successful_steps) == (rs.iterations, rs.termination, rs.num_successful_steps)
    assert np.abs(d["pose"] - ref["pose"]).max() <= POSE_TOL
    assert lmk_err(d["lmk"], ref["lmk"]) <= LMK_TOL
    if "log" in ref:   # live oracle solve: iterate-by-iterate parity, incl. the attempts that follow a rejected step
        assert_trace_matches(trace, ref["log"], rs.termination, cost_rtol=1e-8)
    if vio:
        for k in ("dv", "dba", "dbg"):
            assert np.abs(d[k] - ref[k]).max() <= POSE_TOL
    return s, d, ref


@pytest.mark.parametrize("factor", [capi.FACTOR_PIXEL, capi.FACTOR_ANGULAR])
def test_vo_prior_on_kept_landmarks_lds_path(backend_cls, oracle_lib, factor):
    w = synthetic.make_window(n_kf=6, n_lmk=400, seed=31, factor=factor)
    w.dense_prior = random_prior(w, 12, -1, np.random.default_rng(1))
    s, d, ref = compare(backend_cls, oracle_lib, w, capi.reference_options())
    # the prior matters: without it the solution differs
    plain = oracle_lib.solve(w, capi.reference_options())
    assert np.abs(plain["lmk"] - ref["lmk"]).max() > 1e-4


def test_vio_prior_on_kept_frame_and_landmarks_lds_path(backend_cls, oracle_lib):
    w = make_vio_window(n_kf=6, n_lmk=300, seed=5)
    kf_keep = w.n_kf - 2  # oldest free key-frame (the oldest one is constant)
    w.dense_prior = random_prior(w, 20, kf_keep, np.random.default_rng(2))
    compare(backend_cls, oracle_lib, w, capi.reference_options(), vio=True)


def test_vio_prior_hbm_path_config3_shape(backend_cls, oracle_lib):
    """Config-3 shaped: 12 key-frames, 15 states each, ~300 kept landmarks: N_p = 165 + 900 > 174."""
    w = make_vio_window(n_kf=12, n_lmk=3000, seed=6)
    kf_keep = w.n_kf - 2
    w.dense_prior = random_prior(w, 300, kf_keep, np.random.default_rng(3), rank_deficit=5)
    compare(backend_cls, oracle_lib, w, capi.reference_options(), vio=True, golden="config3_shape_ref_solve")


def test_prior_with_constant_kept_variables(backend_cls, oracle_lib):
    """Kept landmarks / key-frame that are constant in this solve contribute their (zero) delta only."""
    w = synthetic.make_window(n_kf=5, n_lmk=200, seed=33)
    w.dense_prior = random_prior(w, 8, w.n_kf - 1, np.random.default_rng(4))  # kept frame = the constant oldest one
    w.lmk_const = np.zeros(w.n_lmk, dtype=np.uint8)
    w.lmk_const[w.dense_prior["lmk_index"][0]] = 1
    compare(backend_cls, oracle_lib, w, capi.reference_options())


def test_clearing_the_prior_restores_the_plain_solve(backend_cls, oracle_lib):
    w = synthetic.make_window(n_kf=5, n_lmk=200, seed=34)
    dp = random_prior(w, 8, -1, np.random.default_rng(5))
    be = backend_cls(device=0)
    be.set_windows([w])
    base = be.solve(capi.reference_options())[0].final_cost
    be.set_dense_prior(0, dp)
    with_prior = be.solve(capi.reference_options())[0].final_cost
    be.set_dense_prior(0, None)
    again = be.solve(capi.reference_options())[0].final_cost
    be.close()
    assert abs(with_prior - base) > 1e-3 and np.isclose(again, base, rtol=1e-12)


def test_vo_prior_hbm_path(backend_cls, oracle_lib):
    """Pure VO (6 states per key-frame) with 250 kept landmarks: N_p = 774, dense 96-column panels."""
    w = synthetic.make_window(n_kf=5, n_lmk=600, seed=35)
    w.dense_prior = random_prior(w, 250, -1, np.random.default_rng(9), rank_deficit=0)
    compare(backend_cls, oracle_lib, w, capi.reference_options())


@pytest.mark.parametrize("n_keep", [39, 70, 71, 72, 103])
def test_dense_reduced_system_edge_sizes(backend_cls, oracle_lib, n_keep):
    """The wide-panel dense solver (look-ahead panel loop, per-step back-substitution) at N_p = 75 + 3 n_keep = 192 (two panels
    exactly), 285 / 288 / 291 (three short of / exactly / three beyond a multiple of the 96-column panel) and 384: the last
    block's clamped loads sit at the end of the matrix (a fault found on a window at the end of the S allocation)."""
    w = make_vio_window(n_kf=6, n_lmk=900, seed=100 + n_keep)
    w.dense_prior = random_prior(w, n_keep, w.n_kf - 2, np.random.default_rng(n_keep), rank_deficit=3)
    compare(backend_cls, oracle_lib, w, capi.reference_options(), vio=True)


def test_one_launch_per_panel_equals_the_round3_panel_loop(backend_cls, oracle_lib):
    """k_wchol_step (one launch per 96 columns: block substitution against the factor's tiles, M = L^-1 off the chain, out-of-place
    panels) against the round-3 loop it replaced (SADVIO_WD_R3: trsm8 + syrk_la with the explicit inverse) on a 3-panel-and-a-bit
    system: same iterations, same accepted steps, solutions equal far below the parity bar."""
    import os
    w = make_vio_window(n_kf=6, n_lmk=900, seed=171)
    w.dense_prior = random_prior(w, 80, w.n_kf - 2, np.random.default_rng(80), rank_deficit=2)    # N_p = 75 + 240 = 315
    out = {}
    for name, val in (("step", None), ("r3", "1")):
        old = os.environ.pop("SADVIO_WD_R3", None)
        if val is not None:
            os.environ["SADVIO_WD_R3"] = val
        try:
            be = backend_cls(device=0)
            be.set_windows([w])
            s = be.solve(capi.reference_options())[0]
            out[name] = (s, be.get_deltas(0))
            be.close()
        finally:
            os.environ.pop("SADVIO_WD_R3", None)
            if old is not None:
                os.environ["SADVIO_WD_R3"] = old
    (sa, da), (sb, db) = out["step"], out["r3"]
    assert (sa.iterations, sa.termination, sa.num_successful_steps) == (sb.iterations, sb.termination, sb.num_successful_steps)
    assert np.isclose(sa.final_cost, sb.final_cost, rtol=1e-11)
    assert np.abs(da["pose"] - db["pose"]).max() <= 1e-9 and np.abs(da["lmk"] - db["lmk"]).max() <= 1e-8
