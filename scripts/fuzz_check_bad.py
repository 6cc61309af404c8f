#!/usr/bin/env python
"""Re-examine the windows a scripts/gpu_fuzz.py sweep flagged (gpurun_out/<name>_bad.json): device vs oracle, the device against itself
(run-to-run), and the oracle's own sensitivity to a 1-ulp nudge of the measurements (tests/conditioning.py). Usage:
python scripts/fuzz_check_bad.py gpurun_out/r04_fuzz_a1_bad.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sadvio_amd import capi
from oracle import oracle
import conditioning, fuzz_helpers as fz

for b in json.load(open(sys.argv[1])):
    spec = b["spec"]
    w = fz.build_window(spec)
    opts = fz.options(b)
    ref = oracle.solve(w, opts, dense_prior=w.dense_prior)
    runs = []
    for rep in range(3):
        be = capi.Backend(device=0, use_graph=b["use_graph"])
        be.set_windows([w]); s = be.solve(opts)[0]; d = be.get_deltas(0); be.close()
        runs.append((s, d))
    dp = [float(np.abs(d["pose"] - ref["pose"]).max()) for s, d in runs]
    dc = [abs(s.final_cost - ref["summary"].final_cost) / abs(ref["summary"].final_cost) for s, d in runs]
    self_dev = float(np.abs(runs[0][1]["pose"] - runs[1][1]["pose"]).max())
    sp, sc, sl, same = conditioning.oracle_self_sensitivity(lambda: fz.build_window(spec), opts, oracle, ref)
    print(fz.describe(spec), "| device-oracle pose", ["%.1e" % v for v in dp], "cost", ["%.1e" % v for v in dc], "| device run-to-run pose %.1e" % self_dev,
          "| oracle 1-ulp self-sensitivity pose %.1e cost %.1e lmk %.1e same-path %s" % (sp, sc, sl, same), "| it", runs[0][0].iterations, ref["summary"].iterations,
          "radius %.1e" % runs[0][0].final_radius, flush=True)
