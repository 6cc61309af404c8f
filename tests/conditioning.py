"""Conditioning-aware pose comparison for the randomised parity sweep (tests/test_gpu_fuzz.py).

BASELINE.json's pose bar (1e-6) is a statement about directions the data determine. A window can hold directions that only the
LM damping determines — e.g. a key-frame with ONE observation (2 equations, 6 unknowns: a 4-dimensional null space of J^T J):
there the Gauss-Newton gradient is zero in exact arithmetic, every float64 implementation sees its own rounding noise
eps * |J| |r| instead, and the LM step divides that noise by the damping D = diag(J^T J) / radius alone, i.e. multiplies it by
the trust-region radius (1e11 .. 1e16 near convergence). Arbitration of the round-2 sweep's only pose disagreement
(seed 39573273, key-frame 0 has one observation; scripts/fuzz_arbitrate.py): against the long-double twin
the float64 twin is off by 5e-8, the float64 twin with a Schur complement by 4e-6, the C oracle by 1.8e-5, the device by a
similar amount — in that key-frame's null space only; every other key-frame agrees to 2e-12.

So a pose difference d (free key-frames stacked) is judged in the eigenbasis of the damped reduced system the last LM step
solved, S_d = S + D_p (S = Schur complement of the landmarks at the final iterate, D = diag(J^T J) / radius_final):

    |v_i . d|  <=  POSE_TOL * max(1, kappa_i / KAPPA0),     kappa_i = lambda_max / lambda_i

— the full 1e-6 for every direction whose relative stiffness is above 1 / KAPPA0 = 1e-9 (all directions of a well-posed
window: they measure 1e-13 .. 1e-11), and a rounding-amplification allowance eps-like * kappa beyond. The strict bar is tried
first; this bound is only consulted when it fails, and the failing window must ALSO agree on the cost to 1e-8 and on the
iteration count / termination, which pins the determined directions.
"""
import numpy as np

KAPPA0 = 1e9
SELF_K = 8.0


def oracle_self_sensitivity(build, opts, oracle_lib, ref, n_draws=8):
    """How far the ORACLE moves when every measurement of the window is nudged by one unit in the last place (up or down, seeded):
    (max pose difference, relative final-cost difference, max landmark difference relative to max(1 m, the landmark's own delta),
    same iteration count / termination) over `n_draws` draws. `build()`
    returns a fresh copy of the window. A window on which 20 LM iterations amplify a 1-ulp input change to 1e-5 in a pose cannot
    be reproduced better than that by ANY second implementation — measured on the round-3 sweeps (scripts/gpu_fuzz.py, 24
    flagged windows): the device-vs-oracle difference tracks this number within a factor 4 over ten decades
    (1e-15 .. 1e-5), so the sweep accepts SELF_K times it where the fixed bar fails. Eight draws (round 6; three before): the maximum over
    three draws underestimated the spread of the pinned window 234496164 by 2.6 x (2.0e-7 against 5.3e-7 over eight or sixteen draws),
    and the device's run-to-run scatter on it (atomic-add order: 1.6e-6 .. 2.1e-6) then crossed 8 x that estimate on one run in a few."""
    dp = dc = dl = 0.0
    same = True
    for t in range(n_draws):
        w2 = build()
        rng = np.random.default_rng(100 + t)
        m = np.asarray(w2.obs_meas)
        w2.obs_meas = np.where(rng.random(m.shape) < 0.5, np.nextafter(m, np.inf), np.nextafter(m, -np.inf))
        r2 = oracle_lib.solve(w2, opts, dense_prior=getattr(w2, "dense_prior", None))
        dp = max(dp, float(np.abs(r2["pose"] - ref["pose"]).max()))
        dc = max(dc, abs(r2["summary"].final_cost - ref["summary"].final_cost) / max(abs(ref["summary"].final_cost), 1e-300))
        if ref["lmk"].size:
            scale = np.maximum(1.0, np.abs(ref["lmk"]).max(axis=1))
            dl = max(dl, float((np.abs(r2["lmk"] - ref["lmk"]).max(axis=1) / scale).max()))
        same = same and (r2["summary"].iterations, r2["summary"].termination) == (ref["summary"].iterations, ref["summary"].termination)
    return dp, dc, dl, same


def _pose_prior_jtj(w, oracle_lib, pose):
    out = []
    for (k, T, inf) in w.pose_priors:
        r, J = oracle_lib.factor_pose_prior(w.kf_T_f_w[k], T, inf, pose[k])
        out.append((int(k), J.T @ J))
    return out


def damped_reduced_system(w, oracle_lib, pose, lmk, radius):
    """(S_d [6 n_free, 6 n_free], free key-frame indices) of a visual window (pixel / bearing factors + pose priors) at the
    iterate (pose, lmk), landmarks eliminated, LM damping of trust-region radius `radius` included. None for windows with
    other factor families (IMU, dense / sparse priors, lines): the caller then keeps the strict bar."""
    if w.has_imu or getattr(w, "dense_prior", None) is not None or getattr(w, "sparse_priors", None) or getattr(w, "lines", None) is not None:
        return None
    r, Jp, Jl, valid = oracle_lib.linearize(w, pose, lmk)
    free = np.flatnonzero(np.asarray(w.kf_const) == 0)
    col = -np.ones(w.n_kf, dtype=np.int64)
    col[free] = 6 * np.arange(len(free))
    n = 6 * len(free)
    H = np.zeros((n, n))
    lmk_const = np.asarray(w.lmk_const) if getattr(w, "lmk_const", None) is not None else np.zeros(w.n_lmk, dtype=np.uint8)
    kf = np.asarray(w.obs_kf)
    for o in range(w.n_obs):
        c = col[kf[o]]
        if c >= 0:
            H[c:c + 6, c:c + 6] += Jp[o].T @ Jp[o]
    for k, JtJ in _pose_prior_jtj(w, oracle_lib, pose):
        if col[k] >= 0:
            H[col[k]:col[k] + 6, col[k]:col[k] + 6] += JtJ
    d = np.diag(H).copy()
    S = H + np.diag(np.maximum(d, 1e-6 * (1 + np.sqrt(d)) ** 2) / radius)
    ptr = np.asarray(w.lmk_obs_ptr)
    for l in range(w.n_lmk):
        if lmk_const[l] or ptr[l + 1] == ptr[l]:
            continue
        os_ = range(ptr[l], ptr[l + 1])
        M = sum(Jl[o].T @ Jl[o] for o in os_)
        dm = np.diag(M).copy()
        M = M + np.diag(np.maximum(dm, 1e-6 * (1 + np.sqrt(dm)) ** 2) / radius)
        Mi = np.linalg.inv(M)
        E = np.zeros((n, 3))
        for o in os_:
            c = col[kf[o]]
            if c >= 0:
                E[c:c + 6] += Jp[o].T @ Jl[o]
        rows = np.flatnonzero(np.abs(E).max(axis=1) > 0)
        if len(rows):
            Er = E[rows]
            S[np.ix_(rows, rows)] -= Er @ Mi @ Er.T
    return 0.5 * (S + S.T), free


def pose_difference_within_conditioning(w, oracle_lib, ref, pose_a, pose_b, pose_tol):
    """True when pose_a - pose_b passes the eigen-direction bound of the module docstring; also returns a short report."""
    radius = float(ref["log"][-1][2]) if "log" in ref and len(ref["log"]) else 1e16
    sys_ = damped_reduced_system(w, oracle_lib, ref["pose"], ref["lmk"], radius)
    if sys_ is None:
        return False, "no conditioning model for this factor mix"
    S, free = sys_
    lam, V = np.linalg.eigh(S)
    lam = np.maximum(lam, 1e-300)
    d = (np.asarray(pose_a) - np.asarray(pose_b))[free].ravel()
    c = np.abs(V.T @ d)
    kappa = lam.max() / lam
    bound = pose_tol * np.maximum(1.0, kappa / KAPPA0)
    worst = int(np.argmax(c / bound))
    ok = bool(np.all(c <= bound))
    return ok, (f"radius {radius:.1e}, worst direction: |v.d| {c[worst]:.2e} at kappa {kappa[worst]:.1e} (bound {bound[worst]:.1e}); "
                f"{int((kappa > KAPPA0).sum())} of {len(lam)} directions beyond kappa0; largest |v.d| among the determined ones "
                f"{c[kappa <= KAPPA0].max() if (kappa <= KAPPA0).any() else 0.0:.2e}")
