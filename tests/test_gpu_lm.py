"""The throughput kernels of sadvio_amd/csrc/lm_kernels.h (k_build_obs / k_lm_pass: the three-pass path large plain
batches take by themselves) forced onto small windows with SADVIO_LM=1: same LM trace as the oracle, key-frame and landmark
deltas within 1e-6, and the same answer as the latency kernels of kernels.h (SADVIO_LM=0)."""
import os

import numpy as np
import pytest

from sadvio_amd import capi
from sadvio_amd.synthetic import make_window
from golden_util import assert_trace_matches

pytestmark = pytest.mark.gpu
TOL = 1e-6


@pytest.fixture
def lm_env():
    old = os.environ.get("SADVIO_LM")
    yield lambda v: os.environ.__setitem__("SADVIO_LM", v)
    if old is None:
        os.environ.pop("SADVIO_LM", None)
    else:
        os.environ["SADVIO_LM"] = old


def solve(backend_cls, ws, opts, use_graph=False, profile=False):
    be = backend_cls(device=0, use_graph=use_graph, profile_kernels=profile)
    try:
        be.set_windows(ws)
        ss = be.solve(opts)
        out = [(ss[i], be.get_deltas(i), be.get_trace(i)) for i in range(len(ws))]
        names = set(be.kernel_times().keys()) if profile else set()
    finally:
        be.close()
    return out, names


@pytest.mark.parametrize("factor", [capi.FACTOR_PIXEL, capi.FACTOR_ANGULAR])
@pytest.mark.parametrize("use_graph", [False, True])
def test_lm_path_matches_oracle_and_latency_path(backend_cls, oracle_lib, lm_env, factor, use_graph):
    ws = [make_window(n_kf=9, n_lmk=1500, obs_per_lmk=5, seed=77, factor=factor), make_window(n_kf=6, n_lmk=700, obs_per_lmk=4, seed=78, factor=factor)]
    ws[1].lmk_const = np.zeros(ws[1].n_lmk, dtype=np.uint8); ws[1].lmk_const[::7] = 1      # some constant landmarks
    ws[0].kf_const = ws[0].kf_const.copy(); ws[0].kf_const[3] = 1                            # a constant key-frame in the middle
    opts = capi.reference_options()
    lm_env("1")
    fast, _ = solve(backend_cls, ws, opts, use_graph)
    lm_env("0")
    slow, _ = solve(backend_cls, ws, opts, use_graph)
    for w, (s, d, tr), (s0, d0, _) in zip(ws, fast, slow):
        ref = oracle_lib.solve(w, opts)
        rs = ref["summary"]
        assert (s.iterations, s.termination, s.num_successful_steps) == (rs.iterations, rs.termination, rs.num_successful_steps)
        assert np.isclose(s.initial_cost, rs.initial_cost, rtol=1e-11) and np.isclose(s.final_cost, rs.final_cost, rtol=1e-8)
        assert_trace_matches(tr, ref["log"], rs.termination)
        assert np.abs(d["pose"] - ref["pose"]).max() <= TOL and np.abs(d["lmk"] - ref["lmk"]).max() <= TOL
        assert s.iterations == s0.iterations
        assert np.abs(d["pose"] - d0["pose"]).max() <= 1e-9 and np.abs(d["lmk"] - d0["lmk"]).max() <= 1e-9


def test_lm_kernels_are_the_ones_that_ran(backend_cls, lm_env):
    w = make_window(n_kf=6, n_lmk=600, obs_per_lmk=5, seed=5)
    lm_env("1")
    _, names = solve(backend_cls, [w], capi.reference_options(), profile=True)
    assert {"k_lm_pass0", "k_build_obs", "k_lm_pass"} <= names and "k_build" not in names
    lm_env("0")
    _, names = solve(backend_cls, [w], capi.reference_options(), profile=True)
    assert "k_build" in names and "k_lm_pass" not in names


def test_robust_loss_keeps_the_latency_kernels(backend_cls, oracle_lib, lm_env):
    """The throughput kernels carry no loss function: with Huber the batch stays on kernels.h even when forced."""
    w = make_window(n_kf=6, n_lmk=600, obs_per_lmk=5, seed=6)
    opts = capi.reference_options(); opts.huber_a = 1.0
    lm_env("1")
    (res,), names = solve(backend_cls, [w], opts, profile=True)
    assert "k_build" in names and "k_lm_pass" not in names
    ref = oracle_lib.solve(w, opts)
    assert np.abs(res[1]["pose"] - ref["pose"]).max() <= TOL


def test_large_batch_takes_the_throughput_path_by_itself_and_agrees_with_the_latency_path(backend_cls, lm_env):
    """BASELINE config 2 at full size, 9 windows per submission (72 000 landmarks >= the 65 536 threshold): the throughput kernels
    run without being asked, and every window's solve equals the one the latency kernels produce (same iterations, costs to
    1e-12, deltas to 1e-9) - the size-independent property at the size the bench measures."""
    os.environ.pop("SADVIO_LM", None)
    ws = [make_window(seed=20250404 + i) for i in range(3)]
    ws = [ws[i % 3] for i in range(9)]
    opts = capi.reference_options()
    fast, names = solve(backend_cls, ws, opts, profile=True)
    assert {"k_lm_pass0", "k_build_obs", "k_lm_pass"} <= names and "k_build" not in names
    lm_env("0")
    slow, names0 = solve(backend_cls, ws[:3], opts, profile=True)
    assert "k_build" in names0
    for k, (s, d, tr) in enumerate(fast):
        s0, d0, tr0 = slow[k % 3]
        assert (s.iterations, s.termination, s.num_successful_steps) == (s0.iterations, s0.termination, s0.num_successful_steps)
        assert np.isclose(s.final_cost, s0.final_cost, rtol=1e-12)
        assert np.allclose(tr[:, 0], tr0[:, 0], rtol=1e-11)
        assert np.abs(d["pose"] - d0["pose"]).max() <= 1e-9 and np.abs(d["lmk"] - d0["lmk"]).max() <= 1e-8


@pytest.mark.parametrize("huber", [0.0, 1.0])
def test_multi_round_tiles_on_the_latency_kernels(backend_cls, oracle_lib, lm_env, huber):
    """Large batches that cannot take the throughput kernels (robust loss, kept landmarks) run k_build with several rounds of
    landmarks per tile (MFMA accumulators carried across the rounds): forced here with SADVIO_TILE_ROUNDS on a small window."""
    lm_env("0")
    old = os.environ.get("SADVIO_TILE_ROUNDS")
    os.environ["SADVIO_TILE_ROUNDS"] = "3"
    try:
        w = make_window(n_kf=8, n_lmk=1200, obs_per_lmk=5, seed=31)
        opts = capi.reference_options(); opts.huber_a = huber
        (res,), names = solve(backend_cls, [w], opts, profile=True)
    finally:
        if old is None:
            os.environ.pop("SADVIO_TILE_ROUNDS", None)
        else:
            os.environ["SADVIO_TILE_ROUNDS"] = old
    assert "k_build" in names
    ref = oracle_lib.solve(w, opts)
    rs = ref["summary"]
    s, d, tr = res
    assert (s.iterations, s.termination) == (rs.iterations, rs.termination)
    assert_trace_matches(tr, ref["log"], rs.termination)
    assert np.abs(d["pose"] - ref["pose"]).max() <= TOL and np.abs(d["lmk"] - ref["lmk"]).max() <= TOL


def test_relayout_of_a_handle_with_another_tiling(backend_cls, oracle_lib, lm_env):
    """k_build_obs sums a tile's key-frame record over every sub-block slot while k_lm_pass writes one slot per
    work item; the record buffer is grow-only, so a second set_windows on the SAME handle with a different tiling must not see the
    first layout's records in the slots nobody writes any more. First a small batch (32-landmark tiles: one slot per tile, every
    record written), then a batch large enough for 128-landmark tiles (two slots per tile, the second never written) on the same
    handle: every window of the second solve must equal the oracle's."""
    opts = capi.reference_options()
    first = [make_window(n_kf=8, n_lmk=2600, obs_per_lmk=5, seed=300 + i) for i in range(3)]
    w = make_window(n_kf=7, n_lmk=2600, obs_per_lmk=5, seed=310)
    lm_env("1")
    be = backend_cls(device=0, use_graph=False)
    try:
        be.set_windows(first)
        be.solve(opts)
        be.set_windows([w] * 80)                         # 208 000 landmarks: four landmark rounds per tile
        ss = be.solve(opts)
        out = [(ss[i], be.get_deltas(i), be.get_trace(i)) for i in (0, 41, 79)]
    finally:
        be.close()
    ref = oracle_lib.solve(w, opts)
    rs = ref["summary"]
    for s, d, tr in out:
        assert (s.iterations, s.termination) == (rs.iterations, rs.termination)
        assert_trace_matches(tr, ref["log"], rs.termination)
        assert np.abs(d["pose"] - ref["pose"]).max() <= TOL and np.abs(d["lmk"] - ref["lmk"]).max() <= TOL
