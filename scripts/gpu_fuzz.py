#!/usr/bin/env python
"""Randomised parity sweep: windows of random shape (key-frames, landmarks, track lengths, constant masks, priors, factor
type, VIO / VO, batches) solved by the HIP path and by the oracle; reports the worst disagreement. Usage:
  python scripts/gpu_fuzz.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
import numpy as np
from sadvio_amd import capi, synthetic
from oracle import oracle
from vio_helpers import make_vio_window
from test_gpu_prior import random_prior
from sparse_helpers import vio_sparse_priors, vo_sparse_priors

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
oracle.build()
t0 = time.time()
n_cases = 0
worst = {"pose": 0.0, "lmk": 0.0, "cost": 0.0}
bad = []
while time.time() - t0 < budget:
    n_win = int(rng.choice([1, 1, 1, 2, 3]))
    vio = bool(rng.random() < 0.35)
    factor = int(rng.choice([capi.FACTOR_PIXEL, capi.FACTOR_ANGULAR]))
    ws, desc = [], []
    for _ in range(n_win):
        n_kf = int(rng.integers(3, 26)) if not vio else int(rng.integers(3, 12))
        # well-posed problems only: tracks spanning at least two key-frames (>= 3 views) and enough landmarks per key-frame;
        # with 2-view (one stereo pair) tracks every key-frame floats on its own and both solvers follow rounding noise
        n_lmk = int(rng.integers(15 * n_kf, 15 * n_kf + 1500))
        opl = int(rng.integers(3, min(2 * n_kf, 14) + 1)) if n_kf > 1 else 2
        seed = int(rng.integers(1, 1 << 30))
        fixed = int(rng.integers(0, min(3, n_kf)))
        kw = dict(n_kf=n_kf, n_lmk=n_lmk, obs_per_lmk=opl, seed=seed, factor=factor, fixed=fixed, length=float(rng.uniform(2, 12)))
        w = make_vio_window(**kw) if vio else synthetic.make_window(**kw)
        if rng.random() < 0.3:
            w.lmk_const = (rng.random(w.n_lmk) < 0.1).astype(np.uint8)
        if rng.random() < 0.3 and n_kf > 1:
            k = int(rng.integers(0, n_kf)); w.pose_priors.append((k, w.kf_T_f_w[k].copy(), float(rng.uniform(1, 200)) * np.ones(6)))
        if fixed == 0 and not w.pose_priors:
            w.pose_priors.append((n_kf - 1, w.kf_T_f_w[n_kf - 1].copy(), 100.0 * np.ones(6)))
        extra = "plain"
        u = rng.random()
        if u < 0.2 and w.n_lmk > 12:
            w.dense_prior = random_prior(w, int(rng.integers(2, min(40, w.n_lmk - 2))), (n_kf - 2 if (vio and n_kf > 2) else -1), rng)
            extra = "dense"
        elif u < 0.4 and w.n_lmk > 12:
            ls = sorted(rng.choice(w.n_lmk, size=int(rng.integers(2, min(30, w.n_lmk))), replace=False).tolist())
            w.sparse_priors = vio_sparse_priors(w, max(n_kf - 2, 0), ls, rng) if vio else vo_sparse_priors(w, ls, rng)
            extra = "sparse"
        ws.append(w); desc.append(f"kf{n_kf} l{n_lmk} o{opl} f{fixed} {extra} seed{seed}")
    opts = capi.reference_options()
    if rng.random() < 0.25: opts.huber_a = 1.345 ** 0.5
    be = capi.Backend(device=0, use_graph=bool(rng.random() < 0.5))
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
                bad.append((desc[k], "vio" if vio else "vo", factor, opts.huber_a, ep, el, ec, sums[k].iterations, rs.iterations, sums[k].termination, rs.termination))
    except Exception as e:
        bad.append((desc, "EXC", str(e)))
    finally:
        be.close()
    n_cases += n_win
print(f"{n_cases} windows in {time.time()-t0:.1f} s; worst |dpose| {worst['pose']:.2e} |dlmk| {worst['lmk']:.2e} rel cost {worst['cost']:.2e}; {len(bad)} disagreement(s)")
bad_exc = [b for b in bad if len(b) == 3]
for b in bad_exc[:5]:
    print("   EXC", b)
rest = sorted([b for b in bad if len(b) > 3], key=lambda b: -max(b[4], b[6]))
for b in rest[:14]:
    print("  ", b[0], b[1], "factor", b[2], "huber %.2f" % b[3], "dpose %.1e dlmk %.1e dcost %.1e" % (b[4], b[5], b[6]), "it", b[7], b[8], "term", b[9], b[10])
