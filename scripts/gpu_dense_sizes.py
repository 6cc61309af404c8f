#!/usr/bin/env python
"""Dense reduced systems of edge sizes through the wide-panel solver (look-ahead panel loop, per-step back-substitution): N_p a
multiple of the 96-column panel, three short / three beyond it, two panels exactly; GPU against the oracle. python scripts/gpu_dense_sizes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from sadvio_amd import capi
from oracle import oracle
from vio_helpers import make_vio_window
from test_gpu_prior import random_prior
oracle.build()
opts = capi.reference_options()
worst = 0.0
for n_keep in (39, 70, 71, 72, 102, 103, 104, 135, 167, 230):      # N_p = 75 + 3 n_keep: 192, 285, 288, 291, 381, 384, 387, 480, 576, 765
    w = make_vio_window(n_kf=6, n_lmk=900, seed=100 + n_keep)
    w.dense_prior = random_prior(w, n_keep, w.n_kf - 2, np.random.default_rng(n_keep), rank_deficit=3)
    be = capi.Backend(device=0)
    be.set_windows([w]); s = be.solve(opts)[0]; d = be.get_deltas(0); be.close()
    ref = oracle.solve(w, opts, dense_prior=w.dense_prior)
    rs = ref["summary"]
    ep = float(np.abs(d["pose"] - ref["pose"]).max()); el = float(np.abs(d["lmk"] - ref["lmk"]).max())
    ec = abs(s.final_cost - rs.final_cost) / abs(rs.final_cost)
    ok = (s.iterations, s.termination) == (rs.iterations, rs.termination) and ep < 1e-6 and ec < 1e-8
    worst = max(worst, ep)
    print(f"N_p {75 + 3 * n_keep:4d}: it {s.iterations}/{rs.iterations} term {s.termination}/{rs.termination} dpose {ep:.2e} dlmk {el:.2e} dcost {ec:.2e} {'ok' if ok else 'MISMATCH'}", flush=True)
print("worst dpose", worst)
