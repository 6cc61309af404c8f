#!/usr/bin/env python
"""Randomised parity sweep of sadvio_ba_marginalize: VIO / VO windows of random shape, with / without an earlier prior, both
eigenvalue cuts, both forms of the prior, then a SECOND marginalisation that folds the first one's resident prior in (the per-key-frame
chain: unpivoted wide-panel route under the reference cut, the (Ak, bk) the resident prior keeps) — each against the oracle's
J^T J, J^T r0 (information of the prior; |r0|^2 where the rank is not in question). Usage: python scripts/gpu_fuzz_marg.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sadvio_amd import capi, synthetic
from oracle import oracle
from marg_helpers import with_lonely_landmarks
from sadvio_amd.synthetic import pre_marginalize
from vio_helpers import make_vio_window

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
oracle.build()


def info(p):
    J, r0 = p["J"], p["r0"]
    return J.T @ J, J.T @ r0


def rel(a, b):
    Ha, ga = info(a); Hb, gb = info(b)
    s = np.abs(Hb).max()
    return float(np.abs(Ha - Hb).max() / s), float(np.abs(ga - gb).max() / max(np.abs(gb).max(), np.sqrt(s)))


t0 = time.time()
n_cases, worst_H, worst_g, bad = 0, 0.0, 0.0, []
while time.time() - t0 < budget:
    vio = bool(rng.random() < 0.7)
    n_kf = int(rng.integers(4, 9)); n_lmk = int(rng.integers(150, 720)); seed = int(rng.integers(1, 2**31 - 1))
    n_lonely = int(rng.integers(0, 30)); cut = "reference" if rng.random() < 0.6 else "noise_floor"
    has_last = bool(rng.random() < 0.6)
    kf0 = n_kf - 1
    base = make_vio_window(n_kf=n_kf, n_lmk=n_lmk, seed=seed) if vio else synthetic.make_window(n_kf=n_kf, n_lmk=n_lmk, seed=seed)
    w = with_lonely_landmarks(base, kf0, n_lonely)
    keep, marg = pre_marginalize(w, kf0)
    if len(keep) < 4:
        continue
    args = dict(kf_marg=kf0, lmk_marg=marg, lmk_keep=keep, priors=getattr(w, "pose_priors", []))
    if vio:
        imu = [f for f in w.imu_factors if f["kf_i"] == kf0 and f["kf_j"] == kf0 - 1][0]
        args.update(kf_keep=kf0 - 1, marg_has_imu=True, imu=imu)
    if has_last:
        nl = 15 if vio else 6
        r2 = np.random.default_rng(seed + 5)
        args["last"] = {"J": 20.0 * (np.eye(nl) + 0.1 * r2.standard_normal((nl, nl))), "r0": 0.1 * r2.standard_normal(nl), "kf_keep": kf0, "kf_col": 0,
                        "lmk_index": np.zeros(0, dtype=np.int32), "lmk_col": np.zeros(0, dtype=np.int32)}
    desc = f"{'vio' if vio else 'vo'} kf{n_kf} l{n_lmk} seed{seed} lonely{n_lonely} {cut} last{int(has_last)}"
    try:
        o = oracle.marginalize(w, eig_cut=cut, **args)
        be = capi.Backend(device=0)
        be.set_windows([w])
        res = {}
        for form in ("cholesky", "eigen"):
            g = be.marginalize(0, eig_cut=cut, form=form, **args)
            res[form] = g
        # chain: fold the resident (eigen-form) prior's successor — re-marginalise with the Cholesky-form prior resident as `last` over
        # the same kept columns (every kept landmark is still in the window)
        g1 = be.marginalize(0, eig_cut=cut, form="cholesky", readback=False, **args)
        chain_args = dict(args, last={"kf_keep": args.get("kf_keep", -1) if vio else -1, "kf_col": g1["kf_col"], "lmk_index": g1["lmk_index"], "lmk_col": g1["lmk_col"]})
        if vio:
            chain_args["last"]["kf_keep"] = kf0 - 1      # the prior sits on frame1 (kept) and the kept landmarks
        g2 = be.marginalize(0, eig_cut=cut, form="cholesky", **chain_args)
        o_chain_last = dict(chain_args["last"], J=res["eigen"]["J"], r0=res["eigen"]["r0"])
        o2 = oracle.marginalize(w, eig_cut=cut, **dict(chain_args, last=o_chain_last))
        be.close()
    except Exception as e:      # noqa
        bad.append((desc, "exception " + repr(e)[:200]))
        continue
    n_cases += 1
    for name, a, b in (("cholesky", res["cholesky"], o), ("eigen", res["eigen"], o), ("chain", g2, o2)):
        eH, eg = rel(a, b)
        worst_H = max(worst_H, eH); worst_g = max(worst_g, eg)
        rank_ok = abs(a["n_full"] - b["n_full"]) <= (16 if cut == "reference" else max(4, int(0.02 * b["n"])))
        if eH > 1e-7 or eg > 1e-6 or not rank_ok:
            bad.append((desc + " " + name, f"dH {eH:.1e} dg {eg:.1e} n_full {a['n_full']} / {b['n_full']} of {b['n']}"))
print(f"{n_cases} marginalisation cases (x 3 comparisons) in {time.time() - t0:.1f} s; worst rel |J^T J - oracle| {worst_H:.2e}, |J^T r0 - oracle| {worst_g:.2e}; {len(bad)} disagreement(s)")
for d, m in bad[:40]:
    print("  ", d, "|", m)
