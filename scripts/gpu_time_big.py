#!/usr/bin/env python
"""Per-kernel times of the HBM-resident (N_p > 174) path: config 4 and config 5 shaped windows on one GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sadvio_amd import capi, synthetic

which = sys.argv[1] if len(sys.argv) > 1 else "c4"
if which == "c4":
    w = synthetic.make_window(n_kf=100, n_lmk=50000, length=50.0, band=6, seed=4)
else:
    w = synthetic.make_window(n_kf=500, n_lmk=200000, length=250.0, band=6, seed=5)
opts = capi.gn_options(10)
opts.max_num_consecutive_invalid_steps = 1000
be = capi.Backend(device=0, profile_kernels=True)
t = time.perf_counter(); be.set_windows([w]); print(f"set_windows {time.perf_counter()-t:.3f} s")
for _ in range(2): be.solve(opts)
be.set_windows([w])
for _ in range(3): s = be.solve(opts)
kt = be.kernel_times()
print(which, "n_kf", w.n_kf, "n_lmk", w.n_lmk, "n_obs", w.n_obs, {k: round(v["avg_us"], 1) for k, v in kt.items()})
print("   cost", s[0].initial_cost, "->", s[0].final_cost, "iters", s[0].iterations, "ok steps", s[0].num_successful_steps)
be.close()
be = capi.Backend(device=0)
be.set_windows([w])
be.solve(opts)
t = time.perf_counter()
for _ in range(3): be.solve(opts)
dt = (time.perf_counter() - t) / 3
print(f"   wall/solve {dt*1e3:.2f} ms -> {10/dt:.1f} it/s")
