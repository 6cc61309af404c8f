"""GPU parity of the visual-inertial solve (IMUFactor + IMUBiasFactor + PosePriordx + reprojection) against the
oracle, and the reference's own end-to-end VI tests (imu_test.cpp:464-487, 545-568) through the C ABI."""
import numpy as np
import pytest

from imu_helpers import CFG, Chain, arr, factor_dict
from sadvio_amd import capi
from sadvio_amd.synthetic import T12_to_4, T_to_12, exp_so3, inv4
from test_oracle_imu import ACC, GYR, _free_fall_chain, _vio_window
from vio_helpers import make_vio_window

from golden_util import lmk_err

pytestmark = pytest.mark.gpu


def gpu_solve(backend_cls, w, opts):
    be = backend_cls(device=0)
    try:
        be.set_windows([w])
        s = be.solve(opts)[0]
        d = be.get_deltas(0)
    finally:
        be.close()
    return s, d


def assert_match(s, d, ref, tol=1e-6):
    rs = ref["summary"]
    assert (s.iterations, s.termination) == (rs.iterations, rs.termination)
    assert np.isclose(s.initial_cost, rs.initial_cost, rtol=1e-9) and np.isclose(s.final_cost, rs.final_cost, rtol=1e-8)
    for k in ("pose", "dv", "dba", "dbg"):
        assert np.abs(d[k] - ref[k]).max() <= tol, k
    if ref["lmk"].size:
        assert lmk_err(d["lmk"], ref["lmk"]) <= tol


def test_reference_vi_test_pose_recovery(backend_cls, oracle_lib):
    """imu_test.cpp:464-487 through the HIP backend: two key-frames, priors, one IMU factor, no landmark."""
    T_i_f, ch, cur, cfg = _free_fall_chain()
    prior_kf = T_to_12(inv4(T_i_f)); prior_cur = arr(cur.T_f_w).copy()
    err = np.array([0, 0, 0, 0.1, 0.05, -0.01])
    D = np.eye(4); D[:3, :3] = exp_so3(err[:3]); D[:3, 3] = err[3:]
    cur.T_f_w[:] = list(T_to_12(T12_to_4(arr(cur.T_f_w)) @ D))
    cur.v[:] = list(arr(cur.v) + np.array([0.04, 0.02, -0.02]))
    f = factor_dict(1, 0, cur, 1.0, cfg)
    w = _vio_window([cur, ch.kf], [(0, prior_cur, 100 * np.ones(6)), (1, prior_kf, 100 * np.ones(6))], [f])
    opts = capi.reference_options()
    s, d = gpu_solve(backend_cls, w, opts)
    ref = oracle_lib.solve(w, opts)
    assert_match(s, d, ref)
    D = np.eye(4); D[:3, :3] = exp_so3(d["pose"][0][:3]); D[:3, 3] = d["pose"][0][3:]
    T_w_f = inv4(T12_to_4(arr(cur.T_f_w)) @ D)
    assert np.linalg.norm(T_w_f[:3, 3] - np.array([1, 1, 1.5])) < 1e-2            # :482
    assert np.linalg.norm(arr(cur.v) + d["dv"][0] - np.array([0, 0, 1])) < 1e-2     # :483
    assert abs((T_w_f[:3, :3].T @ T_i_f[:3, :3]).trace() - 3) < 1e-5               # :484


def test_reference_vi_test_bias_estimation(backend_cls, oracle_lib):
    """imu_test.cpp:545-568 through the HIP backend."""
    ba, bg = np.array([0.5, 1.0, 1.0]), np.array([0.1, 0.3, 0.1])
    ch = Chain(ACC, GYR, 1e9, ba=ba, bg=bg)
    s1 = ch.step(ACC, GYR, 1.5e9)
    f = factor_dict(1, 0, s1, 0.5)
    I12 = T_to_12(np.eye(4))
    w = _vio_window([s1, ch.kf], [(0, I12, 100 * np.ones(6)), (1, I12, 100 * np.ones(6))], [f])
    s, d = gpu_solve(backend_cls, w, capi.reference_options())
    assert np.linalg.norm(d["dbg"][1]) < 1e-5 and np.linalg.norm(d["dba"][1]) < 1e-5
    assert_match(s, d, oracle_lib.solve(w, capi.reference_options()))


@pytest.mark.parametrize("factor", [capi.FACTOR_PIXEL, capi.FACTOR_ANGULAR])
@pytest.mark.parametrize("mode", ["ref", "gn6"])
def test_vio_window_matches_oracle(backend_cls, oracle_lib, factor, mode):
    w = make_vio_window(n_kf=6, n_lmk=300, seed=5, factor=factor)
    opts = capi.reference_options() if mode == "ref" else capi.gn_options(6)
    s, d = gpu_solve(backend_cls, w, opts)
    assert_match(s, d, oracle_lib.solve(w, opts))


def test_vio_window_config3_shape(backend_cls, oracle_lib):
    """EuRoC-shaped VIO window of the reference's shipped size: 12 KF (config.yaml:34), ~600 features / KF."""
    w = make_vio_window(n_kf=12, n_lmk=2900, seed=11)
    opts = capi.reference_options()
    s, d = gpu_solve(backend_cls, w, opts)
    assert_match(s, d, oracle_lib.solve(w, opts))


def test_vio_constant_frames(backend_cls, oracle_lib):
    w = make_vio_window(n_kf=5, n_lmk=150, seed=9, fixed=2)
    opts = capi.reference_options()
    s, d = gpu_solve(backend_cls, w, opts)
    ref = oracle_lib.solve(w, opts)
    assert_match(s, d, ref)
    assert np.abs(d["dv"][w.kf_const == 1]).max() == 0 and np.abs(d["dba"][w.kf_const == 1]).max() == 0


def test_all_constant_imu_factor_on_a_reused_handle(backend_cls, oracle_lib):
    """An IMU factor between two constant key-frames contributes nothing to the reduced system. Its scratch row may hold
    the entries of a factor of the PREVIOUS window on the same handle (found by scripts/gpu_fuzz.py): they must not leak."""
    opts = capi.reference_options()
    w1 = make_vio_window(n_kf=6, n_lmk=250, seed=21, fixed=0)
    w2 = make_vio_window(n_kf=6, n_lmk=250, seed=22, fixed=2)
    be = backend_cls(device=0)
    try:
        be.set_windows([w1]); be.solve(opts)
        be.set_windows([w2])
        s = be.solve(opts)[0]
        d = be.get_deltas(0)
    finally:
        be.close()
    assert_match(s, d, oracle_lib.solve(w2, opts))


@pytest.mark.parametrize("prior", ["none", "sparsified"])
def test_pose_only_factors_as_extra_workgroups_and_as_kernels(backend_cls, oracle_lib, monkeypatch, prior):
    """IMU pairs and listed sparse-prior factors are evaluated by extra workgroups of k_build / k_backsub (a window or two)
    or by kernels of their own on the same stream (batches; SADVIO_PF_WG forces either): both against the oracle, with
    accepted AND rejected steps in the solve (the linearisation rows are kept per delta buffer)."""
    from sparse_helpers import vio_sparse_priors
    w = make_vio_window(n_kf=7, n_lmk=420, seed=31, lmk_perturb=0.3, rot_perturb_deg=2.0)   # a strongly perturbed start: 8 - 9 rejected steps
    if prior == "sparsified":
        w.sparse_priors = vio_sparse_priors(w, w.n_kf - 2, list(range(0, 60, 2)), np.random.default_rng(8), noise=0.03)
    opts = capi.reference_options()
    ref = oracle_lib.solve(w, opts)
    assert ref["summary"].num_unsuccessful_steps > 0 and ref["summary"].num_successful_steps > 0
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SADVIO_PF_WG", mode)
        s, d = gpu_solve(backend_cls, w, opts)
        assert_match(s, d, ref)
        got[mode] = d
    for k in ("pose", "dv", "dba", "dbg"):
        assert np.abs(got["1"][k] - got["0"][k]).max() <= 1e-9, k


def test_vio_window_on_the_throughput_kernels(backend_cls, oracle_lib, monkeypatch):
    """A VIO window on the throughput kernels (lm_kernels.h; by themselves they take batches of >= 65 536 landmarks, SADVIO_LM=1 forces
    them): the IMU pairs are then linearised by k_pf_eval<false> as a kernel of its own, read the LM decision k_decide took and add
    their entries to the reduced system themselves (imu_pair_lin_wg) - with accepted AND rejected steps in the solve."""
    w = make_vio_window(n_kf=7, n_lmk=420, seed=31, lmk_perturb=0.3, rot_perturb_deg=2.0)
    opts = capi.reference_options()
    ref = oracle_lib.solve(w, opts)
    assert ref["summary"].num_unsuccessful_steps > 0 and ref["summary"].num_successful_steps > 0
    monkeypatch.setenv("SADVIO_LM", "1")
    be = backend_cls(device=0, profile_kernels=True)
    try:
        be.set_windows([w])
        s = be.solve(opts)[0]
        d = be.get_deltas(0)
        names = set(be.kernel_times())
    finally:
        be.close()
    assert {"k_build_obs", "k_lm_pass", "k_pf_lin", "k_pf_cost"} <= names and "k_build" not in names, names
    assert_match(s, d, ref)


def test_batch_of_vio_windows(backend_cls, oracle_lib):
    """Several VIO windows in one submission (more tiles than the extra-workgroup variants are used for)."""
    ws = [make_vio_window(n_kf=8, n_lmk=3000, seed=40 + i) for i in range(9)]
    opts = capi.gn_options(4)
    be = backend_cls(device=0)
    try:
        be.set_windows(ws)
        ss = be.solve(opts)
        for i in (0, 4, 8):
            assert_match(ss[i], be.get_deltas(i), oracle_lib.solve(ws[i], opts))
    finally:
        be.close()
