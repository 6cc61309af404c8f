#!/usr/bin/env python
"""Host-time breakdown of sadvio_ba_set_windows (SADVIO_DEBUG=8192 laps of build_layout) on the config-2 window and the config-3 shaped
VIO window: python scripts/set_windows_laps.py"""
import os, sys, time
os.environ["SADVIO_DEBUG"] = "8192"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from sadvio_amd import capi, synthetic
from vio_helpers import make_vio_window
for name, w in (("config-2", synthetic.make_window(seed=20250404)), ("config-3 vio", make_vio_window(n_kf=12, n_lmk=7200, seed=6))):
    be = capi.Backend(device=0)
    for _ in range(4): be.set_windows([w])
    ts = []
    for _ in range(20):
        t = time.perf_counter(); be.set_windows([w]); ts.append(time.perf_counter() - t)
    sys.stderr.flush()
    print(f"== {name}: set_windows median {sorted(ts)[10]*1e3:.3f} ms, min {min(ts)*1e3:.3f} ms", file=sys.stderr, flush=True)
    be.close()
