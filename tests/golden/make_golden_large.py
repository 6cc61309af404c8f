"""Generates the cached oracle solves of the large configurations (tests/golden/config4_ref_solve.npz,
config3_shape_ref_solve.npz): same seeded generator windows as tests/test_gpu_large.py / test_gpu_prior.py,
solved by the CPU oracle (minutes). Run from the repo root:  python tests/golden/make_golden_large.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle  # noqa: E402
from sadvio_amd import capi, synthetic  # noqa: E402
from golden_util import cached_oracle_solve  # noqa: E402
from vio_helpers import make_vio_window  # noqa: E402
from test_gpu_prior import random_prior  # noqa: E402

oracle.build()
w = synthetic.make_window(n_kf=100, n_lmk=50000, length=50.0, band=6, seed=4)
r = cached_oracle_solve("config4_ref_solve", oracle, w, capi.reference_options(), n_threads=8, write=True)
print("config 4:", r["summary"].iterations, r["summary"].final_cost)
w = make_vio_window(n_kf=12, n_lmk=3000, seed=6)
w.dense_prior = random_prior(w, 300, w.n_kf - 2, np.random.default_rng(3), rank_deficit=5)
r = cached_oracle_solve("config3_shape_ref_solve", oracle, w, capi.reference_options(), dense_prior=w.dense_prior, write=True)
print("config 3 shape:", r["summary"].iterations, r["summary"].final_cost)
