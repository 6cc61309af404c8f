#!/usr/bin/env python
"""Wall time of sadvio_ba_marginalize / sadvio_ba_sparsify on a config-3 shaped window (12 KF VIO, ~300 kept landmarks)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from sadvio_amd import capi
from vio_helpers import make_vio_window
from marg_helpers import with_lonely_landmarks
from sadvio_amd.synthetic import pre_marginalize
w = with_lonely_landmarks(make_vio_window(n_kf=12, n_lmk=7200, seed=6), 11, 40)
keep, marg = pre_marginalize(w, 11)
keep = keep[:int(sys.argv[1]) if len(sys.argv) > 1 else 300]
imu = [f for f in w.imu_factors if f["kf_i"] == 11 and f["kf_j"] == 10][0]
rng = np.random.default_rng(1)
last = {"J": 20.0 * (np.eye(15) + 0.1 * rng.standard_normal((15, 15))), "r0": 0.1 * rng.standard_normal(15), "kf_keep": 11, "kf_col": 0,
        "lmk_index": np.zeros(0, dtype=np.int32), "lmk_col": np.zeros(0, dtype=np.int32)}
be = capi.Backend(device=0)
be.set_windows([w])
for rep in range(3):
    t = time.perf_counter()
    g = be.marginalize(0, 11, marg, keep, kf_keep=10, marg_has_imu=True, imu=imu, priors=w.pose_priors, last=last)
    dt = time.perf_counter() - t
    print(f"marginalize: m={g['m']} n={g['n']} n_full={g['n_full']} sweeps={g['sweeps']}  {dt*1e3:.1f} ms", flush=True)
t = time.perf_counter()
fs = be.sparsify(0, g, vio=True)
print(f"sparsify: {len(fs)} factors  {(time.perf_counter()-t)*1e3:.1f} ms")
be.close()
