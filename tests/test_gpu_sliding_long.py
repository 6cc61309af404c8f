"""A LONG sliding sequence: 25 key-frame steps of the reference's back-end loop (slamBiMonoVIO.cpp:561-614,
slamBiMono.cpp:240-286) over a 34-key-frame synthetic trajectory — marginalize the window's oldest key-frame (folding in the prior the
step before left), [sparsify: sparsifyVIO, or the sparsifyVO chain of marginalization.cpp:410-514], drop the frame, solve the next
window with the prior attached, write the deltas back, slide. (dense | sparsified) x (VIO | VO). Device side: the prior never leaves
the handle (resident, Cholesky form); oracle side: the reference's eigen-form (J, r0) carried on the host. The two sides propagate their
state SEPARATELY, so a disagreement compounds. Reported per run: the per-step termination iteration, the device's marginalisation
routes (unpivoted wide-panel / fell back to the rank-revealing one) and the drift between the two trajectories."""
import copy

import numpy as np
import pytest

from sadvio_amd import capi, synthetic
from test_gpu_sliding import apply_deltas, sub_window
from test_oracle_marg import pre_marginalize
from vio_helpers import make_vio_window

pytestmark = pytest.mark.gpu
N_WIN, N_STEPS, N_KF = 8, 25, 34


def vo_sub_window(W, state, kfs):
    w, keep_l = sub_window(W, state, kfs)
    w.has_imu = 0
    w.kf_vel = w.kf_ba = w.kf_bg = None
    w.imu_factors = []
    return w, keep_l


def apply_vo(state, kfs, keep_l, d):
    fake = dict(d, dv=np.zeros((len(kfs), 3)), dba=np.zeros((len(kfs), 3)), dbg=np.zeros((len(kfs), 3)))
    apply_deltas(state, kfs, keep_l, fake)


def cols_in(w, pr):
    """(lmk_index, lmk_col) of a prior's landmarks in window w (col -1: the landmark left the window)."""
    idx, col = [], []
    for lid, lc in zip(pr["lmk_id"], pr["lmk_col"]):
        j = np.flatnonzero(w.lmk_id == lid)
        idx.append(int(j[0]) if len(j) else 0); col.append(int(lc) if len(j) else -1)
    return np.array(idx, dtype=np.int32), np.array(col, dtype=np.int32)


def run_sequence(backend_cls, oracle_lib, vio, sparsif, cut, n_steps=N_STEPS, run=("dev", "ora"), hook=None, n_win=N_WIN, n_kf=N_KF, n_lmk=1100, length=17.0, resync=None, keep_cap=None,
                 snap=None, replica=None, nudge_seed=None, dev_form="cholesky", state_noise=None, seed=977):
    """run: the sides to propagate ("ora" alone runs on a CPU: scripts/rank_arbiter.py); hook(step, side, w, g, args): called after every
    marginalisation; resync(step, log_row, sides, prior, be) -> bool: called after a step whose prior ranks differ, may align the two
    sides again (see test_25_key_frame_steps); snap(step, side, state, kfs2, result, rank): called after every solve + write-back (golden fixtures of
    one side: scripts/gen_sliding_golden.py); replica(step, w, args, w2, prior_template): oracle side of a dense sequence, called before the
    step's solve with everything a second, perturbed evaluation of the SAME step needs (the oracle's self-sensitivity, same script)."""
    N_WIN_, N_KF_ = n_win, n_kf
    if vio:
        W = make_vio_window(n_kf=N_KF_, n_lmk=n_lmk, seed=seed, obs_per_lmk=6, length=length)
    else:
        W = synthetic.make_window(n_kf=N_KF_, n_lmk=n_lmk, seed=seed, obs_per_lmk=6, length=length)
        W.kf_vel = W.kf_ba = W.kf_bg = np.zeros((N_KF_, 3)); W.imu_factors = []
    if nudge_seed is not None:      # every measurement of the trajectory moved by one unit in the last place (self-sensitivity of a whole sequence)
        rng = np.random.default_rng(nudge_seed)
        m = np.asarray(W.obs_meas)
        W.obs_meas = np.where(rng.random(m.shape) < 0.5, np.nextafter(m, np.inf), np.nextafter(m, -np.inf))
    opts = capi.reference_options()
    init = {"T": W.kf_T_f_w.copy(), "p": W.lmk_p.copy(), "v": np.array(W.kf_vel, dtype=float).copy(), "ba": np.array(W.kf_ba, dtype=float).copy(),
            "bg": np.array(W.kf_bg, dtype=float).copy()}
    if state_noise is not None:     # (seed, eps): the initial landmark estimates moved by eps relative — the size of the state difference two
        rng = np.random.default_rng(state_noise[0])     # implementations carry into a step after a few of them (conditioning of a sequence, scripts/)
        init["p"] = init["p"] * (1.0 + state_noise[1] * rng.standard_normal(init["p"].shape))
    sides = {"dev": copy.deepcopy(init), "ora": copy.deepcopy(init)}
    be = backend_cls(device=0) if "dev" in run else None
    oldest = W.n_kf - 1
    mk_win = sub_window if vio else vo_sub_window
    prior = {"dev": None, "ora": None}
    if vio:   # the prior an initialisation leaves on the oldest key-frame's 15 states (see test_gpu_sliding.py)
        J0 = np.diag(np.concatenate([10.0 * np.ones(6), 5.0 * np.ones(3), 20.0 * np.ones(3), 50.0 * np.ones(3)]))
        first = {"kf_id": int(W.kf_id[oldest]), "kf_col": 0, "lmk_id": [], "lmk_col": []}
        prior = {"dev": dict(first), "ora": dict(first, J=J0, r0=np.zeros(15))}
        if be is not None:
            be.set_prior(J0, np.zeros(15))
    log = []
    for step in range(n_steps):
        kfs = list(range(oldest - N_WIN_ + 1 - step, oldest + 1 - step))      # newest first; the last entry is frame0 of this step
        frame0, frame1 = len(kfs) - 1, len(kfs) - 2
        results, ranks = {}, {}
        for side in run:
            st = sides[side]
            w, keep_l = mk_win(W, st, kfs)
            if step == 0:
                w.pose_priors = [(frame0, W.truth["T_f_w"][kfs[frame0]].copy(), 100.0 * np.ones(6))]
            keep, marg = pre_marginalize(w, frame0)
            if keep_cap is not None:      # the shipped configuration holds ~300 landmarks in the prior (config.yaml:108 features per frame, most of them re-observed)
                keep = keep[:keep_cap]
            pr = prior[side]
            if pr is not None:      # landmarks the previous prior holds are kept if they are still in the window (marginalization.cpp:116-139)
                for lid in pr["lmk_id"]:
                    j = np.flatnonzero(w.lmk_id == lid)
                    if len(j) and int(j[0]) not in keep and int(j[0]) not in marg:
                        keep.append(int(j[0]))
            last = None
            if pr is not None:
                idx, col = cols_in(w, pr)
                last = {"kf_keep": frame0 if vio else -1, "kf_col": pr["kf_col"] if vio else 0, "lmk_index": idx, "lmk_col": col}
                if vio:
                    assert pr["kf_id"] == w.kf_id[frame0]            # the kept frame of the last step is frame0 now
                if side == "ora":
                    last["J"], last["r0"] = pr["J"], pr["r0"]
            args = dict(kf_marg=frame0, lmk_marg=marg, lmk_keep=keep, priors=w.pose_priors, last=last, eig_cut=cut)
            if vio:
                args.update(kf_keep=frame1, marg_has_imu=True, imu=[f for f in w.imu_factors if f["kf_i"] == frame0 and f["kf_j"] == frame1][0])
            if side == "dev":
                be.set_windows([w])
                g = be.marginalize(0, form=dev_form, readback=False, **args)
                fs = be.sparsify(0, g, vio=vio) if sparsif else None
            else:
                g = oracle_lib.marginalize(w, **args)
                fs = oracle_lib.sparsify(w, g, vio=vio) if sparsif else None
            assert g is not None and (not sparsif or fs is not None), (step, side)
            if hook is not None:
                hook(step, side, w, g, args)
            ranks[side] = (int(g["n_full"]), int(g["n"]))
            new_prior = {"kf_id": int(w.kf_id[frame1]), "kf_col": g["kf_col"], "lmk_id": [int(w.lmk_id[l]) for l in keep], "lmk_col": list(g["lmk_col"])}
            if side == "ora":
                new_prior["J"], new_prior["r0"] = g["J"], g["r0"]
            prior[side] = new_prior
            # next window: frame0 dropped (discardLastFrame). VIO: nothing constant, the prior anchors the window; VO: the oldest
            # key-frame is held (fixed_frame_number: 1, config.yaml:35; ...Analytic.cpp:219)
            kfs2 = kfs[:-1]
            w2, keep_l2 = mk_win(W, st, kfs2)
            if not vio:
                w2.kf_const = np.zeros(len(kfs2), dtype=np.uint8); w2.kf_const[-1] = 1
            if sparsif:
                remap = []
                for f in fs:
                    f = dict(f)
                    if f["kf"] >= 0:
                        f["kf"] = int(np.flatnonzero(w2.kf_id == w.kf_id[f["kf"]])[0])
                    gone = False
                    for key in ("lmk0", "lmk1"):
                        if f[key] >= 0:
                            j = np.flatnonzero(w2.lmk_id == w.lmk_id[f[key]])
                            if len(j):
                                f[key] = int(j[0])
                            else:
                                gone = True
                    if not gone:
                        remap.append(f)
                w2.sparse_priors = remap
            else:
                idx, col = cols_in(w2, new_prior)
                dp = {"kf_keep": int(np.flatnonzero(w2.kf_id == new_prior["kf_id"])[0]) if vio else -1, "kf_col": new_prior["kf_col"] if vio else 0,
                      "lmk_index": idx, "lmk_col": col}
                if side == "ora":
                    if replica is not None:
                        replica(step, w, args, w2, dict(dp))
                    dp["J"], dp["r0"] = g["J"], g["r0"]
                w2.dense_prior = dp
            if side == "dev":
                be.set_windows([w2])
                s = be.solve(opts)[0]
                d = be.get_deltas(0)
                results[side] = (s.iterations, s.termination, s.final_cost, d)
            else:
                r = oracle_lib.solve(w2, opts, dense_prior=w2.dense_prior)
                results[side] = (r["summary"].iterations, r["summary"].termination, r["summary"].final_cost, r)
            (apply_deltas if vio else apply_vo)(st, kfs2, keep_l2, results[side][3])
            if snap is not None:
                snap(step, side, st, kfs2, results[side], ranks[side])
        if len(run) < 2:
            (it, term, cost, _), = results.values()
            log.append(dict(step=step, it=(it, it), term=(term, term), cost=cost, rank=ranks[run[0]]))
            continue
        (it_d, term_d, cost_d, dd), (it_o, term_o, cost_o, do) = results["dev"], results["ora"]
        log.append(dict(step=step, it=(it_d, it_o), term=(term_d, term_o), cost_rel=abs(cost_d - cost_o) / cost_o, dpose=float(np.abs(dd["pose"] - do["pose"]).max()),
                        rank_dev=ranks["dev"], rank_ora=ranks["ora"], drift=float(np.abs(sides["dev"]["T"] - sides["ora"]["T"]).max()), resynced=False))
        if resync is not None and ranks["dev"] != ranks["ora"]:
            log[-1]["resynced"] = bool(resync(step, log[-1], sides, prior, be))
    stats = be.marg_stats() if be is not None else None
    if be is not None:
        be.close()
    return log, stats, sides


def check_sequence(log, stats, sides, tag, min_unpivoted):
    print(tag, "termination iteration per step (device):", [r["it"][0] for r in log])
    print(tag, "oracle where it differs:", [(r["step"], r["it"]) for r in log if r["it"][0] != r["it"][1]])
    print(tag, "prior rank (n_full, n) per step where it is deficient:", [(r["step"], r["rank_dev"], r["rank_ora"]) for r in log if r["rank_dev"][0] != r["rank_dev"][1] or r["rank_dev"] != r["rank_ora"]])
    print(tag, f"marginalisation routes on the device: {stats}; worst per-step |dpose| {max(r['dpose'] for r in log):.2e}, cost {max(r['cost_rel'] for r in log):.2e}, "
               f"drift of the two trajectories after {len(log)} steps {log[-1]['drift']:.2e}")
    # STRICT bars on every step: the device decides the prior's rank by the reference's criterion - eigenvalues
    # above 1e-12 (marginalization.cpp:318-342), evaluated with relative accuracy on the trailing pivots of the rank-revealing Cholesky
    # (ba_capi.hip: refine_rank_by_eigenvalue) - and so keeps the same directions as the oracle's eigen-decomposition, also at the two
    # steps of the VIO dense sequence where an eigenvalue lies below the cut (exact values 3e-23 and 1.1e-14:
    # profiles/r06_rank_arbiter.txt). Round 5 ran this test with a loosened bar after step 13 (228 against 227 directions).
    for r in log:
        assert r["rank_dev"] == r["rank_ora"], r
        assert r["it"][0] == r["it"][1] and r["term"][0] == r["term"][1], r
        assert r["cost_rel"] <= 1e-7, r
        assert r["dpose"] <= 1e-6, r
    assert stats["unpivoted"] >= min_unpivoted, stats
    assert log[-1]["drift"] <= 1e-6
    # landmarks: relative to max(1 m, distance) (near-zero parallax sends some of them kilometres away on both sides alike)
    mag = np.maximum(1.0, np.abs(sides["ora"]["p"]).max(axis=1))
    rel = np.abs(sides["dev"]["p"] - sides["ora"]["p"]).max(axis=1) / mag
    worst = int(rel.argmax())
    print(tag, f"landmarks: 99th percentile of the relative difference {np.percentile(rel, 99):.2e}, worst {rel[worst]:.2e} (landmark {worst}, {mag[worst]:.1f} m away, "
               f"{int((rel > 1e-6).sum())} of {len(rel)} above 1e-6)")
    far = mag > 1e3          # runaway landmarks: depth unobservable (near-zero parallax), the optimisation itself sends them kilometres away
    print(tag, f"landmarks within 1 km: worst {rel[~far].max():.2e}; beyond ({int(far.sum())}): worst {rel[far].max() if far.any() else 0.0:.2e}")
    assert np.percentile(rel, 99) <= 1e-6
    assert rel[~far].max() <= 1e-6
    assert not far.any() or rel[far].max() <= 1e-3


@pytest.mark.parametrize("vio,sparsif", [(True, False), (True, True), (False, False), (False, True)])
def test_25_key_frame_steps(backend_cls, oracle_lib, vio, sparsif):
    log, stats, sides = run_sequence(backend_cls, oracle_lib, vio, sparsif, "reference")
    tag = f"[sliding {'VIO' if vio else 'VO'} {'sparsified' if sparsif else 'dense'}]"
    # routes: the unpivoted wide-panel factorisation except for first calls and behind a prior that dropped a direction
    check_sequence(log, stats, sides, tag, len(log) - 4)
    # (the shipped window size - 12 key-frames, n = 915 - runs three such steps against the oracle in tests/test_gpu_sliding.py; the oracle's
    # eigen-decomposition takes minutes per step there, which is what bounds the length of that sequence, not the device)
