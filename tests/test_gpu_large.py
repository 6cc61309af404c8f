"""GPU parity for windows whose reduced system does not fit LDS (N_p > 174): S is built into a full
row-major matrix in HBM, factorised in place by the library's own kernels (band / cyclic-reduction / panel Cholesky,
dense_chol.h), with the same device-side LM control as the small path. BASELINE.json configs 4 (100 KF x 50 k landmarks) and a scaled config 5."""
import functools

import numpy as np
import pytest

from golden_util import assert_trace_matches, cached_oracle_solve, lmk_err
from sadvio_amd import capi, synthetic

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-6
LMK_TOL = 1e-6   # = the pose bar; relative for landmarks that move by more than a metre (golden_util.lmk_err)


def _compare(backend_cls, oracle_lib, w, opts, check_iters=True, n_threads=2, golden=None):
    be = backend_cls(device=0)
    try:
        be.set_windows([w])
        s = be.solve(opts)[0]
        d = be.get_deltas(0)
        ids = be.get_ids(0)
    finally:
        be.close()
    ref = cached_oracle_solve(golden, oracle_lib, w, opts, n_threads=n_threads) if golden else oracle_lib.solve(w, opts, n_threads=n_threads)
    rs = ref["summary"]
    assert np.isclose(s.initial_cost, rs.initial_cost, rtol=1e-10)
    assert np.isclose(s.final_cost, rs.final_cost, rtol=1e-9)
    if check_iters:
        assert (s.iterations, s.termination, s.num_successful_steps) == (rs.iterations, rs.termination, rs.num_successful_steps)
    assert np.abs(d["pose"] - ref["pose"]).max() <= POSE_TOL
    assert lmk_err(d["lmk"], ref["lmk"]) <= LMK_TOL
    assert np.array_equal(ids[0], w.kf_id) and np.array_equal(ids[1], w.lmk_id)
    return s


@pytest.mark.parametrize("factor", [capi.FACTOR_PIXEL, capi.FACTOR_ANGULAR])
def test_medium_window_out_of_lds(backend_cls, oracle_lib, factor):
    """40 key-frames: N_p = 234, the smallest size class on the HBM-resident path."""
    w = synthetic.make_window(n_kf=40, n_lmk=4000, length=20.0, band=5, seed=11, factor=factor)
    assert 6 * int((w.kf_const == 0).sum()) > 174
    s = _compare(backend_cls, oracle_lib, w, capi.reference_options())
    assert s.final_cost < s.initial_cost


def test_config4_100kf_50k_landmarks(backend_cls, oracle_lib):
    """BASELINE.json config 4 on one GPU: 100 KF x 50 000 landmarks x 250 000 factors, N_p = 594."""
    w = synthetic.make_window(n_kf=100, n_lmk=50000, length=50.0, band=6, seed=4)
    assert (w.n_kf, w.n_lmk, w.n_obs) == (100, 50000, 250000)
    _compare(backend_cls, oracle_lib, w, capi.reference_options(), golden="config4_ref_solve")


def test_mixed_batch_small_and_large(backend_cls, oracle_lib):
    """One submission holding an LDS-sized and an HBM-sized window: both solved, neither disturbed."""
    wa = synthetic.make_window(n_kf=6, n_lmk=500, seed=21)
    wb = synthetic.make_window(n_kf=36, n_lmk=2500, length=18.0, band=5, seed=22)
    opts = capi.reference_options()
    be = backend_cls(device=0)
    be.set_windows([wa, wb, wa])
    sums = be.solve(opts)
    for k, w in enumerate([wa, wb, wa]):
        ref = oracle_lib.solve(w, opts)
        d = be.get_deltas(k)
        assert np.isclose(sums[k].final_cost, ref["summary"].final_cost, rtol=1e-9)
        assert sums[k].iterations == ref["summary"].iterations
        assert np.abs(d["pose"] - ref["pose"]).max() <= POSE_TOL
    be.close()


@pytest.mark.parametrize("band", [3, None])
def test_vio_window_out_of_lds(backend_cls, oracle_lib, band):
    """30-key-frame VIO window (15 states per key-frame, N_p = 435): banded (k_band_solve with 5-column pivots) when
    the co-visibility band is narrow, dense 96-column panels otherwise."""
    from vio_helpers import make_vio_window
    kw = dict(length=15.0, band=band) if band else {}
    w = make_vio_window(n_kf=30, n_lmk=1500, seed=17, **kw)
    opts = capi.reference_options()
    be = backend_cls(device=0)
    be.set_windows([w])
    s = be.solve(opts)[0]
    d = be.get_deltas(0)
    be.close()
    ref = oracle_lib.solve(w, opts)
    rs = ref["summary"]
    assert (s.iterations, s.termination) == (rs.iterations, rs.termination)
    assert np.isclose(s.final_cost, rs.final_cost, rtol=1e-9)
    assert np.abs(d["pose"] - ref["pose"]).max() <= POSE_TOL and lmk_err(d["lmk"], ref["lmk"]) <= LMK_TOL
    for q in ("dv", "dba", "dbg"):
        assert np.abs(d[q] - ref[q]).max() <= POSE_TOL


def test_very_long_band_block_cyclic_reduction(backend_cls, oracle_lib, monkeypatch):
    """270 key-frames (N_p = 1614, bw = 66 -> 25 diagonal blocks): the reduced system goes through the block cyclic
    reduction (dense_chol.h); same answer as the twisted band solver and as the oracle's dense Cholesky."""
    w = synthetic.make_window(n_kf=270, n_lmk=14000, length=135.0, band=6, seed=44)
    opts = capi.gn_options(3)

    def run():
        be = backend_cls(device=0)
        try:
            be.set_windows([w])
            s = be.solve(opts)[0]
            return s, be.get_deltas(0)
        finally:
            be.close()

    s_bcr, d_bcr = run()
    monkeypatch.setenv("SADVIO_NO_BCR", "1")
    s_tw, d_tw = run()
    monkeypatch.delenv("SADVIO_NO_BCR")
    ref = oracle_lib.solve(w, opts, n_threads=2)
    for s, d in ((s_bcr, d_bcr), (s_tw, d_tw)):
        assert np.isclose(s.final_cost, ref["summary"].final_cost, rtol=1e-8)
        assert np.abs(d["pose"] - ref["pose"]).max() <= 1e-6 and np.abs(d["lmk"] - ref["lmk"]).max() <= 1e-5
    assert np.abs(d_bcr["pose"] - d_tw["pose"]).max() <= 1e-8


@functools.lru_cache(maxsize=1)
def _config5():
    w = synthetic.make_window(n_kf=500, n_lmk=200000, length=250.0, band=6, seed=5)
    assert (w.n_kf, w.n_lmk, w.n_obs) == (500, 200000, 1000000)
    return w


def _check_config5(s, d_pose, d_lmk, ref, cost_rtol=1e-8):
    rs = ref["summary"]
    assert (s.iterations, s.termination, s.num_successful_steps) == (rs.iterations, rs.termination, rs.num_successful_steps)
    assert np.isclose(s.initial_cost, rs.initial_cost, rtol=1e-10), (s.initial_cost, rs.initial_cost)
    assert np.isclose(s.final_cost, rs.final_cost, rtol=cost_rtol), (s.final_cost, rs.final_cost)
    dp = np.abs(d_pose - ref["pose"]).max()
    assert dp <= POSE_TOL, dp
    sub = d_lmk[:: ref["lmk_stride"]]
    # landmark deltas of this badly initialised 500-key-frame window reach hundreds of metres: LMK_TOL relative to the
    # landmark's own delta beyond 1 m (tests/test_gpu_fuzz.py)
    dl = (np.abs(sub - ref["lmk"]).max(axis=1) / np.maximum(1.0, np.abs(ref["lmk"]).max(axis=1))).max()
    assert dl <= LMK_TOL, dl
    sq = (d_lmk ** 2).sum()
    assert np.isclose(sq, ref["lmk_sq_norm"], rtol=1e-7), (sq, ref["lmk_sq_norm"])   # the landmarks the fixture does not store


def test_config5_500kf_200k_landmarks_full_size(backend_cls, oracle_lib):
    """BASELINE.json config 5 at FULL size on one GPU: 500 KF x 200 000 landmarks x 1 000 000 factors (N_p = 2 994, block
    cyclic reduction), reference options (20 LM iterations, 7 of them rejected), against the committed oracle solve
    (tests/golden/config5_ref_solve.npz, 84 s of CPU; every 8th landmark + the norm of all of them), iterate by iterate."""
    w = _config5()
    opts = capi.reference_options()
    ref = cached_oracle_solve("config5_ref_solve", oracle_lib, w, opts, n_threads=8)
    assert "lmk_stride" in ref, "config-5 fixture missing or stale (tests/golden/make_golden_large.py)"
    be = backend_cls(device=0)
    try:
        be.set_windows([w])
        s = be.solve(opts)[0]
        d = be.get_deltas(0)
        trace = be.get_trace(0)
    finally:
        be.close()
    _check_config5(s, d["pose"], d["lmk"], ref)
    assert_trace_matches(trace, ref["log"], ref["summary"].termination, cost_rtol=1e-8)


def test_config5_sharded_8_way(backend_cls, oracle_lib):
    """The same window landmark-sharded over 8 ranks (8 handles / streams on the one GPU of the box, host-mediated
    all-reduce of the band-packed reduced system): what `bench.py --shard-window` runs on an 8-GPU node."""
    from test_gpu_sharded import solve_sharded
    w = _config5()
    opts = capi.reference_options()
    ref = cached_oracle_solve("config5_ref_solve", oracle_lib, w, opts, n_threads=8)
    out, coll = solve_sharded(backend_cls, w, opts, 8)
    lmk = np.concatenate([o[1]["lmk"] for o in out])
    assert np.array_equal(np.concatenate([o[2][1] for o in out]), w.lmk_id)
    # 20 LM iterations of a badly initialised 500-key-frame window amplify the rounding differences of a different
    # summation order (8 partial sums instead of one) more than the single-device run does: cost to 1e-7
    for s, d, _ in out:
        _check_config5(s, d["pose"], lmk, ref, cost_rtol=1e-7)
    for r in range(1, 8):
        assert np.array_equal(out[r][1]["pose"], out[0][1]["pose"])
    assert coll.max_count < 2994 * 2994 // 8      # only the band travels
