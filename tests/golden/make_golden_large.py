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

# config 5 at full size: every 8th landmark delta + the squared norm of all of them (4.8 MB of incompressible doubles otherwise)
w = synthetic.make_window(n_kf=500, n_lmk=200000, length=250.0, band=6, seed=5)
opts = capi.reference_options()
r = oracle.solve(w, opts, n_threads=8)
s = r["summary"]
from golden_util import _window_checksum  # noqa: E402
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "config5_ref_solve.npz"), checksum=_window_checksum(w), iterations=s.iterations,
                    termination=s.termination, num_successful_steps=s.num_successful_steps, initial_cost=s.initial_cost,
                    final_cost=s.final_cost, pose=r["pose"], lmk_stride=8, lmk=r["lmk"][::8], lmk_sq_norm=float((r["lmk"] ** 2).sum()),
                    dv=np.zeros((0, 3)), dba=np.zeros((0, 3)), dbg=np.zeros((0, 3)), log=r["log"])
print("config 5:", s.iterations, s.final_cost)
