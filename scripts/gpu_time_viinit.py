#!/usr/bin/env python
"""Wall time of sadvio_ba_vi_init (the whole 50-iteration LM solve in one kernel launch) on a 10-key-frame map."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from sadvio_amd import capi
from viinit_helpers import make_viinit
be = capi.Backend(device=0)
for n in (10, 48):
    pb = make_viinit(n_kf=n, scale=0.5, tilt=(0.05, -0.08), vel_noise=0.01)
    for _ in range(3): r = be.vi_init(pb["T_f_w"], pb["vel"], pb["factors"], optim_scale=True)
    t = time.perf_counter()
    for _ in range(20): r = be.vi_init(pb["T_f_w"], pb["vel"], pb["factors"], optim_scale=True)
    print(f"VIInit {n} key-frames: {(time.perf_counter()-t)/20*1e3:.3f} ms per call, {r['summary'].iterations} iterations, scale {r['scale']:.5f}")
be.close()
