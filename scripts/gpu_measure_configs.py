#!/usr/bin/env python
"""One table of BA iterations/s for the BASELINE.json configurations on one MI355X (GN-10 solves, inputs resident):
config 2 at batch sizes 1 / 8 / 64 / 512 (SURVEY.md §8d iii), a config-3 shaped VIO window (12 KF, IMU factors, dense
prior with 300 kept landmarks / the same prior sparsified), config 4, config 5."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from sadvio_amd import capi, synthetic

opts = capi.gn_options(10)
opts.max_num_consecutive_invalid_steps = 1000
rows = []


def run(name, ws, reps):
    be = capi.Backend(device=0, use_graph=True)
    be.set_windows(ws)
    for _ in range(2): be.solve(opts)
    t = time.perf_counter()
    for _ in range(reps): s = be.solve(opts)
    dt = (time.perf_counter() - t) / reps
    be.close()
    its = sum(x.iterations for x in s)
    n_obs = sum(w.n_obs for w in ws); n_lmk = sum(w.n_lmk for w in ws)
    n_p = [6 * int((w.kf_const == 0).sum()) * (15 // 6 if False else 1) for w in ws]
    b_iter = 120 * n_obs + 96 * n_lmk
    rows.append({"config": name, "windows": len(ws), "ms_per_solve": round(1e3 * dt, 3), "it_per_s": round(its / dt, 1),
                 "algorithmic_GBps_obs_lmk": round(its * b_iter / len(ws) / dt / 1e9 * 1.0, 1) if False else round(10 * b_iter / dt / 1e9, 1)})
    print(rows[-1], flush=True)


base = [synthetic.make_window(seed=20250404 + i) for i in range(8)]
for B in (1, 8, 64, 512):
    run("config 2 (20 KF x 8k lmk x 40k obs)", [base[i % 8] for i in range(B)], 20 if B <= 64 else 3)

from vio_helpers import make_vio_window
from test_gpu_prior import random_prior
from sparse_helpers import vio_sparse_priors
w = make_vio_window(n_kf=12, n_lmk=7200, seed=6)                       # ~600 features per key-frame
run("config 3 shape: 12-KF VIO window, IMU factors, no prior", [w], 20)
w.dense_prior = random_prior(w, 300, w.n_kf - 2, np.random.default_rng(3), rank_deficit=5)
run("config 3 shape + dense prior (300 kept landmarks, N_p = 1065)", [w], 5)
w.dense_prior = None
w.sparse_priors = vio_sparse_priors(w, w.n_kf - 2, list(range(0, 600, 2)), np.random.default_rng(4), noise=0.03)
run("config 3 shape + sparsified prior (300 pose-to-landmark factors)", [w], 5)
run("config 4 (100 KF x 50k lmk x 250k obs), one GPU", [synthetic.make_window(n_kf=100, n_lmk=50000, length=50.0, band=6, seed=4)], 5)
run("config 5 (500 KF x 200k lmk x 1M obs), one GPU", [synthetic.make_window(n_kf=500, n_lmk=200000, length=250.0, band=6, seed=5)], 3)
json.dump(rows, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "configs_table.json"), "w"), indent=1)
