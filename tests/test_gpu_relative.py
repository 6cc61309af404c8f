"""SURVEY.md §8f rank 2 on the GPU: Relative6DPose factors in the window solve, sadvio_ba_marginalize_relative, and pose
graphs over recovered factors — against the oracle (tests/test_oracle_relative.py pins that side on the reference's
formulas). Includes config 5 read as "a global problem built from sparsified window priors": the 500-key-frame window's
consecutive pairs reduced to relative-pose factors (sadvio_ba_marginalize_relative) and solved as one pose graph."""
import numpy as np
import pytest

from sadvio_amd import capi, synthetic
from test_oracle_relative import loop_graph, perturbed, rel_window, relative_prior, compose

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-6


def solve_both(backend_cls, oracle_lib, g, opts, use_graph=False):
    be = backend_cls(device=0, use_graph=use_graph)
    try:
        be.set_windows([g])
        s = be.solve(opts)[0]
        d = be.get_deltas(0)
        trace = be.get_trace(0)
    finally:
        be.close()
    ref = oracle_lib.solve(g, opts)
    rs = ref["summary"]
    assert (s.iterations, s.termination, s.num_successful_steps) == (rs.iterations, rs.termination, rs.num_successful_steps)
    assert np.isclose(s.initial_cost, rs.initial_cost, rtol=1e-10)
    assert abs(s.final_cost - rs.final_cost) <= 1e-8 * rs.final_cost + 1e-14 * rs.initial_cost
    assert np.abs(d["pose"] - ref["pose"]).max() <= POSE_TOL
    return s, d, ref, trace


@pytest.mark.parametrize("use_graph", [False, True])
def test_pose_graph_matches_oracle_and_recovers_the_poses(backend_cls, oracle_lib, use_graph):
    rng = np.random.default_rng(8)
    Twf, factors = loop_graph(12, rng)                 # N_p = 66: the in-LDS solver
    pert = perturbed(Twf, rng)
    g = rel_window(pert, factors, fixed=(0,))
    opts = capi.reference_options(); opts.max_num_iterations = 50; opts.function_tolerance = 1e-14
    s, d, ref, _ = solve_both(backend_cls, oracle_lib, g, opts, use_graph)
    for k in range(len(Twf)):
        assert np.abs(compose(pert[k], d["pose"][k]) - synthetic.T12_to_4(Twf[k])).max() < 1e-7
    # free gauge + pose prior, reference options
    g2 = rel_window(pert, factors, fixed=(), priors=[(0, Twf[0], 100.0 * np.ones(6))])
    solve_both(backend_cls, oracle_lib, g2, capi.reference_options(), use_graph)


def test_relative_factors_next_to_visual_factors(backend_cls, oracle_lib):
    """The factor rides any window (here: a visual window, the slots read as the factor's T_a / T_b)."""
    rng = np.random.default_rng(3)
    w = synthetic.make_window(n_kf=7, n_lmk=500, seed=12)
    for a, b in ((0, 1), (2, 5), (6, 3)):
        W = np.diag(rng.uniform(20, 60, 6)) + rng.standard_normal((6, 6))
        w.sparse_priors.append(dict(type=capi.SPARSE_RELATIVE_POSE, kf=a, kf_b=b, T_prior=relative_prior(w.kf_T_f_w[a], w.kf_T_f_w[b]), sqrt_inf=W))
    opts = capi.reference_options()
    be = backend_cls(device=0)
    be.set_windows([w])
    s = be.solve(opts)[0]
    d = be.get_deltas(0)
    be.close()
    ref = oracle_lib.solve(w, opts)
    assert (s.iterations, s.termination) == (ref["summary"].iterations, ref["summary"].termination)
    assert np.isclose(s.final_cost, ref["summary"].final_cost, rtol=1e-9)
    assert np.abs(d["pose"] - ref["pose"]).max() <= POSE_TOL and np.abs(d["lmk"] - ref["lmk"]).max() <= 1e-5


@pytest.mark.parametrize("factor", [capi.FACTOR_PIXEL, capi.FACTOR_ANGULAR])
def test_marginalize_relative_matches_oracle(backend_cls, oracle_lib, factor):
    w = synthetic.make_window(n_kf=5, n_lmk=900, obs_per_lmk=6, seed=14, factor=factor)
    be = backend_cls(device=0)
    be.set_windows([w])
    for a, b in ((0, 1), (2, 4), (3, 1)):
        got = be.marginalize_relative(0, a, b)
        ref = oracle_lib.marginalize_relative(w, a, b)
        assert (got is None) == (ref is None)
        if ref is None:
            continue
        inf, Ak = got
        assert np.abs(Ak - ref[1]).max() <= 1e-9 * np.abs(ref[1]).max()
        assert np.abs(inf - ref[0]).max() <= 1e-7 * np.abs(ref[0]).max()
    be.close()
    w2 = synthetic.make_window(n_kf=12, n_lmk=60, obs_per_lmk=3, seed=2, band=1, length=40.0)
    be = backend_cls(device=0)
    be.set_windows([w2])
    assert be.marginalize_relative(0, 0, 11) is None           # nothing shared: refused
    be.close()


def test_config5_as_a_pose_graph_over_recovered_factors(backend_cls, oracle_lib):
    """BASELINE.json config 5 ("global BA 500 KF + factor-graph sparsification pass") as the §8f problem: every consecutive
    key-frame pair (and every pair two apart) of a 500-key-frame window is reduced to a Relative6DPose factor whose
    information comes from sadvio_ba_marginalize_relative on the visual window; the 500-node pose graph (N_p = 2 994, block
    banded) is then solved by the window solver and by the oracle."""
    w = synthetic.make_window(n_kf=500, n_lmk=40000, length=250.0, band=6, seed=5, pixel_noise=0.5)
    Twf = [synthetic.T_to_12(synthetic.inv4(synthetic.T12_to_4(T))) for T in w.kf_T_f_w]
    be = backend_cls(device=0)
    be.set_windows([w])
    factors, checked = [], 0
    for k in range(w.n_kf - 1):
        for step in (1, 2):
            b = k + step
            if b >= w.n_kf:
                continue
            got = be.marginalize_relative(0, k, b)
            if got is None:
                continue
            inf = 0.5 * (got[0] + got[0].T)
            if k % 97 == 0:                               # spot-check the information against the oracle
                ref = oracle_lib.marginalize_relative(w, k, b)
                assert np.abs(got[1] - ref[1]).max() <= 1e-9 * np.abs(ref[1]).max()
                checked += 1
            ev, V = np.linalg.eigh(inf)
            Wm = (V * np.sqrt(np.maximum(ev, 1e-6 * ev.max()))) @ V.T        # symmetric square root, weak directions floored
            factors.append(dict(type=capi.SPARSE_RELATIVE_POSE, kf=k, kf_b=b, T_prior=relative_prior(Twf[k], Twf[b]), sqrt_inf=Wm / np.sqrt(ev.max()) * 30.0))
    be.close()
    assert checked >= 5 and len(factors) > 900
    pert = perturbed(Twf, np.random.default_rng(6), rot=0.01, trans=0.05, skip=(w.n_kf - 1,))
    g = rel_window(pert, factors, fixed=(w.n_kf - 1,))
    opts = capi.gn_options(6)
    s, d, ref, trace = solve_both(backend_cls, oracle_lib, g, opts)
    assert s.final_cost < 1e-6 * s.initial_cost
    from golden_util import assert_trace_matches
    assert_trace_matches(trace, ref["log"], ref["summary"].termination, cost_rtol=1e-7)
