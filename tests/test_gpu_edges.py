"""Edge cases of the window layout through the C ABI: maximum track lengths (every tile mode of k_build), windows
without landmarks, mixed batches on the kernels that carry the rare paths, capacity errors."""
import numpy as np
import pytest

from sadvio_amd import capi, synthetic
from sparse_helpers import vio_sparse_priors
from test_gpu_prior import random_prior
from vio_helpers import make_vio_window

from golden_util import lmk_err

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-6
LMK_TOL = 1e-6   # = the pose bar; relative for landmarks that move by more than a metre (golden_util.lmk_err)


def agree(be, k, w, ref, s, vio=False):
    d = be.get_deltas(k)
    rs = ref["summary"]
    assert (s.iterations, s.termination) == (rs.iterations, rs.termination)
    assert np.isclose(s.final_cost, rs.final_cost, rtol=1e-9)
    assert np.abs(d["pose"] - ref["pose"]).max() <= POSE_TOL
    assert d["lmk"].size == 0 or lmk_err(d["lmk"], ref["lmk"]) <= LMK_TOL
    if vio:
        for q in ("dv", "dba", "dbg"):
            assert np.abs(d[q] - ref[q]).max() <= POSE_TOL


@pytest.mark.parametrize("obs_per_lmk", [12, 40, 64])
def test_long_tracks_use_every_tile_mode(backend_cls, oracle_lib, obs_per_lmk):
    """12 views: LDS tiles with ds_add_f64; 40 / 64 views (= MAX_LMK_OBS, one wave per landmark): more than 20 free
    key-frames per tile -> global-atomics tiles."""
    w = synthetic.make_window(n_kf=34, n_lmk=120, obs_per_lmk=obs_per_lmk, seed=91, length=6.0)
    assert np.diff(w.lmk_obs_ptr).max() == obs_per_lmk
    opts = capi.reference_options()
    be = backend_cls(device=0)
    be.set_windows([w])
    s = be.solve(opts)[0]
    agree(be, 0, w, oracle_lib.solve(w, opts), s)
    be.close()


def test_capacity_errors(backend_cls):
    w = synthetic.make_window(n_kf=34, n_lmk=50, obs_per_lmk=64, seed=92, length=6.0)
    # a 65th observation of landmark 0
    w.obs_kf = np.insert(w.obs_kf, 0, w.obs_kf[0]); w.obs_cam = np.insert(w.obs_cam, 0, w.obs_cam[0])
    w.obs_meas = np.insert(w.obs_meas, 0, w.obs_meas[0], axis=0)
    w.lmk_obs_ptr = w.lmk_obs_ptr.copy(); w.lmk_obs_ptr[1:] += 1
    be = backend_cls(device=0)
    with pytest.raises(capi.SadvioError, match="64 observations"):
        be.set_windows([w])
    w = synthetic.make_window(n_kf=3, n_lmk=20, seed=1)
    w.cam_K = np.tile(w.cam_K, (5, 1)); w.cam_T_s_f = np.tile(w.cam_T_s_f, (5, 1)); w.cam_sigma = np.tile(w.cam_sigma, 5)
    be.set_windows([w])                      # ten table entries, two distinct cameras: stored once each
    w.cam_K = w.cam_K + 1e-3 * np.arange(10)[:, None]
    with pytest.raises(capi.SadvioError, match="8 distinct cameras"):
        be.set_windows([w])
    be.close()


def test_window_without_landmarks_inertial_only(backend_cls, oracle_lib):
    """No visual factor at all: pose priors + IMU chain (the reduced system is only built by the solve kernel)."""
    w = make_vio_window(n_kf=5, n_lmk=40, seed=93)
    w.lmk_p = np.zeros((0, 3)); w.lmk_obs_ptr = np.zeros(1, dtype=np.int32); w.lmk_id = np.zeros(0, dtype=np.int64)
    w.obs_kf = np.zeros(0, dtype=np.int32); w.obs_cam = np.zeros(0, dtype=np.int32); w.obs_meas = np.zeros((0, 2))
    w._keep = []
    for k in range(w.n_kf - 1):  # anchor every frame weakly, else the problem is rank deficient
        w.pose_priors.append((k, w.kf_T_f_w[k].copy(), 3.0 * np.ones(6)))
    opts = capi.reference_options()
    be = backend_cls(device=0)
    be.set_windows([w])
    s = be.solve(opts)[0]
    agree(be, 0, w, oracle_lib.solve(w, opts), s, vio=True)
    be.close()


def test_mixed_vio_batch_on_the_extras_kernels(backend_cls, oracle_lib):
    """One submission: a plain VIO window, one with a dense prior, one with sparse prior factors."""
    wa = make_vio_window(n_kf=5, n_lmk=200, seed=94)
    wb = make_vio_window(n_kf=6, n_lmk=250, seed=95)
    wb.dense_prior = random_prior(wb, 15, wb.n_kf - 2, np.random.default_rng(6))
    wc = make_vio_window(n_kf=5, n_lmk=200, seed=96)
    wc.sparse_priors = vio_sparse_priors(wc, wc.n_kf - 2, list(range(0, 30, 2)), np.random.default_rng(7), noise=0.05)
    opts = capi.reference_options()
    be = backend_cls(device=0)
    be.set_windows([wa, wb, wc])
    sums = be.solve(opts)
    agree(be, 0, wa, oracle_lib.solve(wa, opts), sums[0], vio=True)
    agree(be, 1, wb, oracle_lib.solve(wb, opts, dense_prior=wb.dense_prior), sums[1], vio=True)
    agree(be, 2, wc, oracle_lib.solve(wc, opts), sums[2], vio=True)
    be.close()


def test_huber_loss_with_kept_landmarks(backend_cls, oracle_lib):
    """Robust loss and prior-kept landmarks together (k_build_kept applies the same corrector)."""
    from frontend_helpers import with_outliers
    w = with_outliers(synthetic.make_window(n_kf=6, n_lmk=300, seed=97), frac=0.08, seed=5)
    w.dense_prior = random_prior(w, 15, -1, np.random.default_rng(8))
    opts = capi.reference_options(); opts.huber_a = 1.345 ** 0.5
    be = backend_cls(device=0)
    be.set_windows([w])
    s = be.solve(opts)[0]
    agree(be, 0, w, oracle_lib.solve(w, opts, dense_prior=w.dense_prior), s)
    be.close()


def test_one_camera_entry_per_keyframe_like_the_reference(backend_cls, oracle_lib):
    """SaDVIO has one ImageSensor per (frame, camera): 2 N_kf table entries, all copies of the rig's two cameras."""
    w = synthetic.make_window(n_kf=6, n_lmk=300, seed=98)
    n_kf = w.n_kf
    cam_of = w.obs_cam.copy()
    w.obs_cam = (2 * w.obs_kf + cam_of).astype(np.int32)          # entry index = 2 * key-frame + camera
    w.cam_K = np.tile(w.cam_K, (n_kf, 1)); w.cam_T_s_f = np.tile(w.cam_T_s_f, (n_kf, 1)); w.cam_sigma = np.tile(w.cam_sigma, n_kf)
    opts = capi.reference_options()
    be = backend_cls(device=0)
    be.set_windows([w])
    s = be.solve(opts)[0]
    agree(be, 0, w, oracle_lib.solve(w, opts), s)
    r, Jp, Jl = be.linearize(0)
    ro, Jpo, Jlo, _ = oracle_lib.linearize(w)
    assert np.abs(r - ro).max() <= 1e-10 * max(1.0, np.abs(ro).max())
    be.close()


@pytest.mark.parametrize("what", ["nan_measurement", "inf_landmark"])
def test_non_finite_input_fails_cleanly(backend_cls, oracle_lib, what):
    """A NaN measurement / an infinite landmark: every step is invalid, the solve ends as Ceres' FAILURE after
    max_num_consecutive_invalid_steps (SADVIO_E_NOT_USABLE, AOptimizer.cpp:259 semantics), deltas stay zero, nothing hangs,
    and the handle solves a clean window right afterwards."""
    w = synthetic.make_window(n_kf=4, n_lmk=100, seed=3)
    if what == "nan_measurement":
        w.obs_meas = w.obs_meas.copy(); w.obs_meas[17, 0] = np.nan
    else:
        w.lmk_p = w.lmk_p.copy(); w.lmk_p[5] = np.inf
    ref = oracle_lib.solve(w, capi.reference_options())
    be = backend_cls(device=0)
    try:
        be.set_windows([w])
        s = be.solve(capi.reference_options())[0]
        d = be.get_deltas(0)
        assert (s.termination, s.iterations) == (ref["summary"].termination, ref["summary"].iterations) == (5, 5)
        assert np.abs(d["pose"]).max() == 0.0 and np.abs(d["lmk"]).max() == 0.0
        good = synthetic.make_window(n_kf=4, n_lmk=100, seed=3)
        be.set_windows([good])
        s2 = be.solve(capi.reference_options())[0]
        assert s2.termination != 5 and np.isfinite(s2.final_cost) and s2.final_cost < s2.initial_cost
    finally:
        be.close()


@pytest.mark.parametrize("graph", [False, True])
def test_handle_reuse_after_every_way_a_solve_can_end(backend_cls, oracle_lib, graph):
    """The reduced system in HBM is an accumulator that k_build adds into and only the back-substitution tiles re-zero
    (zero_s_slice); the pose priors ride linearisation records written behind the previous factorisation. Whatever way a solve
    ends — function tolerance (the last attempt is not applied), iteration limit, gradient tolerance inside k_solve, zero
    iterations — the NEXT solve on the same handle must start from a clean accumulator and fresh records: each solve of the
    sequence equals the oracle's, and the sequence's last solve equals the same solve on a fresh handle to rounding."""
    w = synthetic.make_window(n_kf=8, n_lmk=900, seed=77)
    w.pose_priors.append((3, w.kf_T_f_w[3].copy(), 25.0 * np.ones(6)))
    ref_opts = capi.reference_options()
    gn = capi.gn_options(6)
    grad = capi.reference_options(); grad.gradient_tolerance = 8e3      # ends inside k_solve (gradient tolerance) after four iterations
    zero = capi.gn_options(0)
    seq = [ref_opts, gn, grad, zero, ref_opts, gn]
    be = backend_cls(device=0, use_graph=graph)
    be.set_windows([w])
    for o in seq:
        s = be.solve(o)[0]
        ref = oracle_lib.solve(w, o)
        assert (s.iterations, s.termination) == (ref["summary"].iterations, ref["summary"].termination)
        if ref["summary"].iterations:
            assert np.isclose(s.final_cost, ref["summary"].final_cost, rtol=1e-9)
        assert np.abs(be.get_deltas(0)["pose"] - ref["pose"]).max() <= POSE_TOL
    last = be.get_deltas(0)
    be.close()
    fresh = backend_cls(device=0, use_graph=graph)
    fresh.set_windows([w])
    fresh.solve(gn)
    d = fresh.get_deltas(0)
    fresh.close()
    # (not bit for bit: the tiles' global atomics into S land in a different order every run)
    assert np.abs(last["pose"] - d["pose"]).max() <= 1e-11 and np.abs(last["lmk"] - d["lmk"]).max() <= 1e-9


def test_bracket_with_factor_setters_only_rebuilds(backend_cls, oracle_lib):
    """begin_update / commit_update around factor setters alone (no set_windows in the bracket) must rebuild and upload —
    the next solve runs with the new pose prior, not with the stale device copy."""
    w = synthetic.make_window(n_kf=6, n_lmk=300, seed=17)
    opts = capi.reference_options()
    be = backend_cls(device=0)
    be.set_windows([w])
    s0 = be.solve(opts)[0]
    # a different prior on the fixed key-frame's neighbour: changes the solution
    kf, T, inf = w.pose_priors[0]
    w2 = synthetic.make_window(n_kf=6, n_lmk=300, seed=17)
    w2.pose_priors = [(kf, T, inf), (0, w2.truth["T_f_w"][0].copy(), 1e4 * np.ones(6))]
    pc = w2.priors_c()
    be._check(be.lib.sadvio_ba_begin_update(be.h), "begin_update")
    be._check(be.lib.sadvio_ba_set_pose_priors(be.h, 0, pc[1], pc[0]), "set_pose_priors")
    be._check(be.lib.sadvio_ba_commit_update(be.h), "commit_update")
    s1 = be.solve(opts)[0]
    ref = oracle_lib.solve(w2, opts)
    assert abs(s1.final_cost - s0.final_cost) > 1e-6 * s0.final_cost          # the new factor is in the program
    agree(be, 0, w2, ref, s1)
    be.close()
