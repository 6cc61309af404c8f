"""Elimination numerics of the oracle (CPU): on the two windows of the sweep that needed a long-double arbiter (tests/golden/fuzz_seed*_ld.npz,
scripts/fuzz_arbitrate.py) the C oracle — landmark elimination in Cholesky form since round 5 — must land within north_star's 1e-6 of
the arbiter. With the adjugate inverse of rounds 1 - 4 it landed 1.8e-5 / 2.1e-4 away (scripts/elim_numerics.py); the
float64 twin shows the three linear solves side by side on the smaller window."""
import json
import os

import numpy as np
import pytest

import fuzz_helpers as fz
from oracle import oracle, twin

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _case(seed):
    pins = json.load(open(os.path.join(GOLDEN, "fuzz_pinned.json")))
    b = [b for b in pins if b["spec"]["seed"] == seed][0]
    return fz.build_window(b["spec"]), fz.options(b), np.load(os.path.join(GOLDEN, f"fuzz_seed{seed}_ld.npz"))


@pytest.mark.parametrize("seed,bar", [(39573273, 1e-6), (961174670, 1e-7)])
def test_oracle_against_the_long_double_arbiter(seed, bar):
    w, opts, z = _case(seed)
    ref = oracle.solve(w, opts, dense_prior=w.dense_prior)
    assert np.abs(ref["pose"] - z["pose"]).max() <= bar


def test_cholesky_blocks_beat_the_adjugate_inverse_in_the_twin():
    """Same factors, same LM schedule, float64; only the landmark elimination differs (19 key-frames, 309 landmarks: seconds)."""
    w, opts, z = _case(39573273)
    err = {}
    for elim in ("cholesky", "adjugate"):
        r = twin.lm_solve(w, opts, kind="f64", use_schur=True, elim=elim)
        err[elim] = float(np.abs(np.asarray(r["pose"], dtype=np.float64) - z["pose"]).max())
    assert err["cholesky"] <= 1e-6 < err["adjugate"], err
