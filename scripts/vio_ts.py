#!/usr/bin/env python
"""In-kernel phase timestamps of k_solve on the config-3 shaped VIO window (TS build of the library, SADVIO_DEBUG=4096):
SADVIO_BA_LIB=<ts build> python scripts/vio_ts.py [none|sparse|dense]"""
import os, sys
os.environ["SADVIO_DEBUG"] = "4096"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from sadvio_amd import capi
from vio_helpers import make_vio_window
from sparse_helpers import vio_sparse_priors
which = sys.argv[1] if len(sys.argv) > 1 else "none"
opts = capi.gn_options(10); opts.max_num_consecutive_invalid_steps = 1000
w = make_vio_window(n_kf=12, n_lmk=7200, seed=6)
if which == "sparse":
    w.sparse_priors = vio_sparse_priors(w, w.n_kf - 2, list(range(0, 600, 2)), np.random.default_rng(4), noise=0.03)
if which == "dense":
    from test_gpu_prior import random_prior
    w.dense_prior = random_prior(w, 300, w.n_kf - 2, np.random.default_rng(3), rank_deficit=5)
be = capi.Backend(device=0)
be.set_windows([w])
for _ in range(3):
    be.solve(opts)
be.close()
