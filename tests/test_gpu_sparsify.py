"""GPU parity of the NFR sparsification (sadvio_ba_sparsify; sparsifyVIO / sparsifyVO, marginalization.cpp:362-514) and
of the whole device pipeline marginalize -> sparsify -> set_sparse_priors -> solve against the oracle."""
import numpy as np
import pytest

from marg_helpers import with_lonely_landmarks
from sadvio_amd import capi
from test_oracle_sparsify import vio_prior, vo_prior
from vio_helpers import make_vio_window

pytestmark = pytest.mark.gpu


def same_factors(fg, fo, rtol=1e-7):
    assert fg is not None and fo is not None and len(fg) == len(fo)
    for a, b in zip(fg, fo):
        assert (a["type"], a["kf"], a["lmk0"], a["lmk1"]) == (b["type"], b["kf"], b["lmk0"], b["lmk1"])
        assert np.allclose(a["delta"], b["delta"], rtol=1e-12, atol=1e-12)
        Wg, Wo = a["sqrt_inf"], b["sqrt_inf"]
        assert np.abs(Wg - Wo).max() <= rtol * np.abs(Wo).max()
        if a["type"] == capi.SPARSE_IMU_PRIOR:
            for k in ("T_prior", "v_prior", "ba_prior", "bg_prior"):
                assert np.array_equal(a[k], b[k])


def test_sparsify_vio(backend_cls, oracle_lib):
    w, pr = vio_prior(oracle_lib)
    be = backend_cls(device=0)
    be.set_windows([w])
    fg = be.sparsify(0, pr, vio=True)
    be.close()
    same_factors(fg, oracle_lib.sparsify(w, pr, vio=True))


@pytest.mark.parametrize("factor", [capi.FACTOR_PIXEL, capi.FACTOR_ANGULAR])
def test_sparsify_vo_chain(backend_cls, oracle_lib, factor):
    w, pr = vo_prior(oracle_lib, factor=factor)
    be = backend_cls(device=0)
    be.set_windows([w])
    fg = be.sparsify(0, pr, vio=False)
    be.close()
    same_factors(fg, oracle_lib.sparsify(w, pr, vio=False))


def test_device_pipeline_marginalize_sparsify_solve(backend_cls, oracle_lib):
    seed = 84
    w = with_lonely_landmarks(make_vio_window(n_kf=6, n_lmk=300, seed=seed), 5, 8)
    from test_oracle_marg import pre_marginalize
    keep, marg = pre_marginalize(w, 5)
    imu = [f for f in w.imu_factors if f["kf_i"] == 5 and f["kf_j"] == 4][0]
    rng = np.random.default_rng(seed)
    last = {"J": 20.0 * (np.eye(15) + 0.1 * rng.standard_normal((15, 15))), "r0": 0.1 * rng.standard_normal(15), "kf_keep": 5,
            "kf_col": 0, "lmk_index": np.zeros(0, dtype=np.int32), "lmk_col": np.zeros(0, dtype=np.int32)}
    args = dict(kf_marg=5, lmk_marg=marg, lmk_keep=keep, kf_keep=4, marg_has_imu=True, imu=imu, priors=w.pose_priors, last=last)
    be = backend_cls(device=0)
    be.set_windows([w])
    pg = be.marginalize(0, **args)
    fg = be.sparsify(0, pg, vio=True)
    po = oracle_lib.marginalize(w, **args)
    fo = oracle_lib.sparsify(w, po, vio=True)
    assert len(fg) == len(fo)
    for a, b in zip(fg, fo):   # the two priors agree to ~1e-8, the information square roots inherit that
        assert np.abs(a["sqrt_inf"] - b["sqrt_inf"]).max() <= 1e-5 * np.abs(b["sqrt_inf"]).max()

    def next_window(fs):
        w2 = with_lonely_landmarks(make_vio_window(n_kf=6, n_lmk=300, seed=seed), 5, 8)
        w2.pose_priors = []; w2.kf_const = np.zeros(w2.n_kf, dtype=np.uint8); w2.kf_const[5] = 1
        w2.imu_factors = [f for f in w2.imu_factors if f["kf_i"] != 5]
        w2.sparse_priors = fs
        return w2

    opts = capi.reference_options()
    be.set_windows([next_window(fg)])
    s = be.solve(opts)[0]
    d = be.get_deltas(0)
    be.close()
    ref = oracle_lib.solve(next_window(fo), opts)
    assert s.iterations == ref["summary"].iterations
    assert np.isclose(s.final_cost, ref["summary"].final_cost, rtol=1e-6)
    assert np.abs(d["pose"] - ref["pose"]).max() <= 1e-6 and np.abs(d["lmk"] - ref["lmk"]).max() <= 1e-5
