#!/usr/bin/env python
"""Elimination numerics (parity depth I): how far from a LONG-DOUBLE solve of the un-reduced system do float64 solves land
that differ only in how the landmarks are eliminated?
  (i)   un-reduced normal equations, LAPACK Cholesky (no elimination; what the float64 twin does)
  (ii)  Schur complement with the landmark blocks taken through a 3 x 3 CHOLESKY — the block step of a landmark-first Cholesky of the
        un-reduced system, i.e. what CHOLMOD does under a fill-reducing ordering (ceres SPARSE_NORMAL_CHOLESKY, AOptimizer.cpp:315-323)
  (iii) Schur complement with the ADJUGATE 3 x 3 inverse and an explicitly formed S -= (E M^-1) E^T — today's device / oracle arithmetic
all in oracle/twin.py (same factors, same LM schedule), + the C oracle itself. CPU only.
  python scripts/elim_numerics.py pinned        the two arbitrated windows of tests/golden (their long-double fixtures)
  python scripts/elim_numerics.py random [n]    n small ill-conditioned draws (3-view landmarks, short baselines), long-double arbiter computed here"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fuzz_helpers as fz
from oracle import oracle, twin

GOLDEN = os.path.join(ROOT, "tests", "golden")


def row(tag, w, opts, z_pose):
    out = {}
    ref = oracle.solve(w, opts, dense_prior=w.dense_prior)
    out["oracle_C"] = float(np.abs(ref["pose"] - z_pose).max())
    for name, kw in (("unreduced_lapack", dict()), ("schur_cholesky3", dict(use_schur=True, elim="cholesky")), ("schur_adjugate", dict(use_schur=True, elim="adjugate"))):
        r = twin.lm_solve(w, opts, kind="f64", **kw)
        out[name] = float(np.abs(np.asarray(r["pose"], dtype=np.float64) - z_pose).max())
        out[name + "_it"] = int(r["iterations"])
    print(tag, " ".join(f"{k}={v:.2e}" if isinstance(v, float) else f"{k}={v}" for k, v in out.items()), flush=True)
    return out


def pinned():
    pins = json.load(open(os.path.join(GOLDEN, "fuzz_pinned.json")))
    res = {}
    for seed in (39573273, 961174670):
        b = [b for b in pins if b["spec"]["seed"] == seed][0]
        w = fz.build_window(b["spec"]); opts = fz.options(b)
        z = np.load(os.path.join(GOLDEN, f"fuzz_seed{seed}_ld.npz"))
        res[seed] = row(f"seed {seed} ({fz.describe(b['spec'])}):", w, opts, z["pose"])
    return res


def random_draws(n):
    rng = np.random.default_rng(20260928)
    res = []
    while len(res) < n:
        case = fz.draw_case(rng)
        spec = case["specs"][0]
        if spec["vio"] or spec["extra"] != "plain" or spec["factor"] != 0:
            continue
        spec = dict(spec, n_kf=min(spec["n_kf"], 8), n_lmk=min(spec["n_lmk"], 90), obs_per_lmk=3, length=min(spec["length"], 1.5))   # short baseline, 3 views: ill-conditioned depth
        w = fz.build_window(spec); opts = fz.options(dict(case, huber=False))
        t = time.time()
        z = twin.lm_solve(w, opts, kind="ld")
        res.append(dict(spec=spec, ld_seconds=time.time() - t, **row(f"draw {len(res)} ({fz.describe(spec)}, arbiter {time.time() - t:.0f} s):", w, opts, np.asarray(z["pose"], dtype=np.float64))))
    return res


if __name__ == "__main__":
    oracle.build()
    mode = sys.argv[1] if len(sys.argv) > 1 else "pinned"
    out = pinned() if mode == "pinned" else random_draws(int(sys.argv[2]) if len(sys.argv) > 2 else 20)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"elim_numerics_{mode}.json"), "w"), indent=1, default=str)
