#!/usr/bin/env python
"""Golden fixture of the ORACLE side of a config-3-size sliding sequence (tests/golden/sliding_config3_size_*.npz): 12-key-frame VIO windows,
~ 600 landmarks per key-frame, 300 landmarks kept in the prior per step (config.yaml:34,108), the reference's eigenvalue cut, N key-frame
steps of marginalize -> [sparsify] -> solve -> write-back (tests/test_gpu_sliding_long.py::run_sequence, run = ("ora",)). The oracle needs
minutes per step at this size (cyclic Jacobi of a 915-column prior, dense-prior solves in plain C), which is why the -m gpu suite cannot
run it beside the device: it is generated once, on the CPU, and the device's sequence is compared with it step by step
(tests/test_gpu_sliding_full_size.py).

Dense variant, optional (third argument = number of draws; the committed fixture was written with 0): beside every step the oracle
evaluates the SAME step again (marginalize + solve from the same state) with every measurement
of the two windows AND every entry of the previous prior (J, r0) nudged by one unit in the last place (N_DRAWS seeded draws, worker
processes): how far the oracle itself moves —
tests/conditioning.py's self-sensitivity, per step. A step behind a rank-deficient prior can amplify a 1-ulp input change to 1e-6 in a pose;
no second implementation reproduces the oracle better than that there, and the test's bar follows that number where the fixed one fails.
Every fixture written with draws also holds the per-iteration log of each solve (layout of sadvio_ba_get_trace).
Usage: python scripts/gen_sliding_golden.py [n_steps] [dense|sparsified] [n_draws] [trajectory seed]"""
import multiprocessing, os, sys, time
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

FULL = dict(n_win=12, n_kf=40, n_lmk=10200, length=20.0, keep_cap=300)
N_DRAWS = int(sys.argv[3]) if len(sys.argv) > 3 else 0
SEED = int(sys.argv[4]) if len(sys.argv) > 4 else 977        # the trajectory (tests/test_gpu_sliding_long.py::run_sequence)


def _nudge(win, rng):
    m = np.asarray(win.obs_meas)
    win.obs_meas = np.where(rng.random(m.shape) < 0.5, np.nextafter(m, np.inf), np.nextafter(m, -np.inf))


def replica_job(payload):
    """One perturbed evaluation of a step (worker process): returns what the baseline's snap records."""
    step, draw, w, args, w2, dpt = payload
    from oracle import oracle
    from sadvio_amd import capi
    rng = np.random.default_rng(1000 * step + draw)
    _nudge(w, rng); _nudge(w2, rng)
    last = args.get("last")
    if last is not None and last.get("J") is not None:      # the previous prior is a rounded quantity too: its entries move by one unit in the last place
        last = dict(last)                                   # (|J|^2 ~ 1e8: this is the eps |Ak| level at which the reference's own eigen-decomposition is accurate)
        for key in ("J", "r0"):
            v = np.asarray(last[key], dtype=float)
            last[key] = np.where(rng.random(v.shape) < 0.5, np.nextafter(v, np.inf), np.nextafter(v, -np.inf))
        args = dict(args, last=last)
    g = oracle.marginalize(w, **args)
    dp = dict(dpt, J=g["J"], r0=g["r0"])
    w2.dense_prior = dp
    r = oracle.solve(w2, capi.reference_options(), dense_prior=dp)
    return step, draw, np.array(r["pose"], dtype=float), float(r["summary"].final_cost), int(r["summary"].iterations), int(r["summary"].termination), (int(g["n_full"]), int(g["n"]))


if __name__ == "__main__":
    from oracle import oracle
    from test_gpu_sliding_long import run_sequence
    n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    sparsif = (sys.argv[2] if len(sys.argv) > 2 else "sparsified") == "sparsified"
    rec = {"it": [], "term": [], "cost": [], "rank": [], "pose": [], "T": [], "kfs": [], "log": []}
    t0 = time.time()
    pool = None if (sparsif or N_DRAWS == 0) else ProcessPoolExecutor(max_workers=5, mp_context=multiprocessing.get_context("spawn"))
    futures = []

    def snap(step, side, st, kfs2, result, rank):
        it, term, cost, r = result
        rec["it"].append(it); rec["term"].append(term); rec["cost"].append(cost); rec["rank"].append(rank)
        rec["pose"].append(np.array(r["pose"], dtype=float).copy()); rec["T"].append(st["T"].copy()); rec["kfs"].append(np.array(kfs2))
        lg = np.zeros((64, 8)); lg[: len(r["log"])] = r["log"]; rec["log"].append(lg)      # the per-iteration log of the solve (same layout as sadvio_ba_get_trace)
        print(f"step {step}: it {it} term {term} cost {cost:.8f} rank {rank}  ({time.time() - t0:.0f} s)", flush=True)

    def replica(step, w, args, w2, dpt):
        for draw in range(N_DRAWS):
            futures.append(pool.submit(replica_job, (step, draw, w, args, w2, dpt)))

    log, _, sides = run_sequence(None, oracle, True, sparsif, "reference", n_steps=n_steps, run=("ora",), snap=snap, replica=None if pool is None else replica, seed=SEED, **FULL)
    extra = {}
    if pool is not None:
        sens_pose = np.zeros(n_steps); sens_cost = np.zeros(n_steps); sens_same = np.ones(n_steps, dtype=np.uint8)
        for f in futures:
            step, draw, pose, cost, it, term, rank = f.result()
            dp = float(np.abs(pose - rec["pose"][step]).max()); dc = abs(cost - rec["cost"][step]) / rec["cost"][step]
            same = (it, term, tuple(rank)) == (rec["it"][step], rec["term"][step], tuple(rec["rank"][step]))
            print(f"  replica step {step} draw {draw}: |dpose| {dp:.2e} cost {dc:.2e} it {it}/{rec['it'][step]} term {term}/{rec['term'][step]} rank {rank}", flush=True)
            sens_pose[step] = max(sens_pose[step], dp); sens_cost[step] = max(sens_cost[step], dc); sens_same[step] &= np.uint8(same)
        pool.shutdown()
        extra = dict(sens_pose=sens_pose, sens_cost=sens_cost, sens_same=sens_same, sens_draws=N_DRAWS)
    out = os.path.join(ROOT, "tests", "golden", f"sliding_config3_size_{'sparsified' if sparsif else 'dense'}{'' if SEED == 977 else '_s%d' % SEED}.npz")
    np.savez_compressed(out, it=np.array(rec["it"]), term=np.array(rec["term"]), cost=np.array(rec["cost"]), rank=np.array(rec["rank"]),
                        pose=np.array(rec["pose"]), T=np.array(rec["T"]), kfs=np.array(rec["kfs"]), p=sides["ora"]["p"], log=np.array(rec["log"]),
                        params=np.array([FULL["n_win"], FULL["n_kf"], FULL["n_lmk"], FULL["keep_cap"], n_steps]), length=FULL["length"], seed=SEED, **extra)
    print("wrote", out, os.path.getsize(out), "bytes")
