#!/usr/bin/env python
"""Randomised parity sweep: windows of random shape (key-frames, landmarks, track lengths, constant masks, priors, factor
type, VIO / VO, batches) solved by the HIP path and by the oracle; reports the worst disagreement and writes the specs of
the disagreeing windows (tests/fuzz_helpers.py format) + both solutions to gpurun_out/fuzz_bad.{json,npz}, from where
they go into tests/test_gpu_fuzz.py as pinned cases. Usage: python scripts/gpu_fuzz.py [seconds] [seed]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
import numpy as np
from sadvio_amd import capi
from oracle import oracle
import fuzz_helpers as fz

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
oracle.build()
t0 = time.time()
n_cases = 0
worst = {"pose": 0.0, "lmk": 0.0, "cost": 0.0}
bad, dump = [], {}
while time.time() - t0 < budget:
    case = fz.draw_case(rng)
    ws = [fz.build_window(s) for s in case["specs"]]
    opts = fz.options(case)
    be = capi.Backend(device=0, use_graph=case["use_graph"])
    try:
        be.set_windows(ws)
        sums = be.solve(opts)
        for k, w in enumerate(ws):
            d = be.get_deltas(k)
            ref = oracle.solve(w, opts, dense_prior=w.dense_prior)
            rs = ref["summary"]
            ep = float(np.abs(d["pose"] - ref["pose"]).max()); el = float(np.abs(d["lmk"] - ref["lmk"]).max()) if w.n_lmk else 0.0
            ec = abs(sums[k].final_cost - rs.final_cost) / max(abs(rs.final_cost), 1e-300)
            same = (sums[k].iterations, sums[k].termination) == (rs.iterations, rs.termination)
            worst["pose"] = max(worst["pose"], ep); worst["lmk"] = max(worst["lmk"], el); worst["cost"] = max(worst["cost"], ec)
            if ep > 1e-6 or el > 1e-5 or ec > 1e-8 or not same:
                i = len(bad)
                bad.append(dict(spec=case["specs"][k], huber=case["huber"], use_graph=case["use_graph"], n_win=len(ws), dpose=ep, dlmk=el, dcost=ec,
                                it=[int(sums[k].iterations), int(rs.iterations)], term=[int(sums[k].termination), int(rs.termination)]))
                if i < 24:
                    dump[f"gpu_pose_{i}"] = d["pose"]; dump[f"gpu_lmk_{i}"] = d["lmk"]; dump[f"ref_pose_{i}"] = ref["pose"]; dump[f"ref_lmk_{i}"] = ref["lmk"]
    except Exception as e:
        bad.append(dict(spec=case["specs"], exc=str(e)))
    finally:
        be.close()
    n_cases += len(ws)
print(f"{n_cases} windows in {time.time()-t0:.1f} s; worst |dpose| {worst['pose']:.2e} |dlmk| {worst['lmk']:.2e} rel cost {worst['cost']:.2e}; {len(bad)} disagreement(s)")
for b in sorted([b for b in bad if "exc" not in b], key=lambda b: -max(b["dpose"], b["dcost"], 0.1 * b["dlmk"]))[:14]:
    print("  ", fz.describe(b["spec"]), "huber" if b["huber"] else "", "dpose %.1e dlmk %.1e dcost %.1e" % (b["dpose"], b["dlmk"], b["dcost"]), "it", b["it"], "term", b["term"])
for b in [b for b in bad if "exc" in b][:5]:
    print("   EXC", b)
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
json.dump(bad, open(os.path.join(out, "fuzz_bad.json"), "w"), indent=1)
if dump:
    np.savez_compressed(os.path.join(out, "fuzz_bad.npz"), **dump)
