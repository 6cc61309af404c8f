#!/usr/bin/env python
"""Wall time (upload + solve + read-back, i.e. what the front-end thread waits for) of the front-end solves of
AOptimizer.cpp:98-297 on windows of the reference's shipped size (~600 features per frame, config.yaml:108)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from sadvio_amd import capi
from frontend_helpers import landmark_optimization_window, single_frame_window, with_outliers
from vio_helpers import make_vio_window

cases = {
    "landmarkOptimization (5 KF const, 600 lmk, Huber, 10 it)": (landmark_optimization_window(n_kf=5, n_lmk=600), capi.landmark_optimization_options(), True),
    "singleFrameOptimization (1 frame, 600 lmk const, 5 it)": (single_frame_window(n_lmk=600), capi.single_frame_options(), False),
}
w = with_outliers(make_vio_window(n_kf=2, n_lmk=600, seed=55, fixed=0, obs_per_lmk=4), frac=0.05, seed=3)
w.lmk_const = np.ones(w.n_lmk, dtype=np.uint8); w.pose_priors = []
cases["singleFrameVIOptimization (2 frames + IMU, 600 lmk const, Huber, 5 it)"] = (w, capi.single_frame_options(vi=True), False)
be = capi.Backend(device=0, use_graph=True)
for name, (w, opts, chi2) in cases.items():
    for _ in range(5):
        be.set_windows([w]); be.solve(opts); be.get_deltas(0)
    t = time.perf_counter(); n = 50
    for _ in range(n):
        be.set_windows([w]); s = be.solve(opts)[0]; d = be.get_deltas(0)
        if chi2: be.landmark_chi2(0)
    dt = (time.perf_counter() - t) / n
    t = time.perf_counter()
    for _ in range(n): s = be.solve(opts)[0]
    ds = (time.perf_counter() - t) / n
    print(f"{name}: {dt*1e3:.3f} ms per call incl. upload / read-back{' / chi2 gate' if chi2 else ''}; solve alone {ds*1e3:.3f} ms, {s.iterations} iterations", flush=True)
be.close()
