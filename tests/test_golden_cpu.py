"""The committed golden vectors are reproduced by the oracle (regression lock of the pinned restatement)."""
import numpy as np
import pytest

from golden_util import load_window
from sadvio_amd import capi


@pytest.mark.parametrize("name", ["window_pixel_5kf", "window_angular_5kf"])
def test_oracle_reproduces_golden(oracle_lib, name):
    w, g = load_window(name)
    r, Jp, Jl, _ = oracle_lib.linearize(w)
    assert np.allclose(r, g["lin0_r"], rtol=1e-12, atol=1e-12) and np.allclose(Jp, g["lin0_Jp"], rtol=1e-12, atol=1e-12)
    r, Jp, Jl, _ = oracle_lib.linearize(w, g["lin_pose_delta"], g["lin_lmk_delta"])
    assert np.allclose(r, g["lin1_r"], rtol=1e-12, atol=1e-12) and np.allclose(Jl, g["lin1_Jl"], rtol=1e-12, atol=1e-12)
    for tag, opts in (("ref", capi.reference_options()), ("gn5", capi.gn_options(5))):
        res = oracle_lib.solve(w, opts)
        s = res["summary"]
        gs = g[f"{tag}_summary"]
        assert (s.iterations, s.num_successful_steps, s.termination) == (int(gs[0]), int(gs[1]), int(gs[3]))
        assert np.isclose(s.final_cost, gs[5], rtol=1e-9)
        assert np.abs(res["pose"] - g[f"{tag}_pose"]).max() < 1e-9
        assert np.abs(res["lmk"] - g[f"{tag}_lmk"]).max() < 1e-8
