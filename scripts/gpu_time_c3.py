#!/usr/bin/env python
"""Per-kernel times of the config-3 shaped VIO window (12 KF, IMU factors) without prior / with a dense prior / with
the sparsified prior."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from sadvio_amd import capi
from vio_helpers import make_vio_window
from test_gpu_prior import random_prior
from sparse_helpers import vio_sparse_priors
opts = capi.gn_options(10); opts.max_num_consecutive_invalid_steps = 1000
which = sys.argv[1] if len(sys.argv) > 1 else "all"
w = make_vio_window(n_kf=12, n_lmk=7200, seed=6)
cases = {"none": (None, [])}
cases["dense"] = (random_prior(w, 300, w.n_kf - 2, np.random.default_rng(3), rank_deficit=5), [])
cases["sparse"] = (None, vio_sparse_priors(w, w.n_kf - 2, list(range(0, 600, 2)), np.random.default_rng(4), noise=0.03))
for name, (dp, sp) in cases.items():
    if which not in ("all", name): continue
    w.dense_prior, w.sparse_priors = dp, sp
    be = capi.Backend(device=0, profile_kernels=True)
    be.set_windows([w])
    for _ in range(2): be.solve(opts)
    be.set_windows([w])
    for _ in range(3): s = be.solve(opts)
    print(name, {k: round(v["avg_us"], 1) for k, v in be.kernel_times().items()}, "cost", s[0].initial_cost, "->", s[0].final_cost, flush=True)
    be.close()
    for ug in (False, True):
        be = capi.Backend(device=0, use_graph=ug)
        be.set_windows([w])
        for _ in range(3): be.solve(opts)
        t = time.perf_counter()
        for _ in range(10): s = be.solve(opts)
        dt = (time.perf_counter() - t) / 10
        print(f"   {name}: graph={ug} wall/solve {dt*1e3:.3f} ms -> {10/dt:.0f} it/s  final cost {s[0].final_cost}", flush=True)
        be.close()
