"""SURVEY.md §8f rank 2 on the CPU: Relative6DPose (residuals.hpp:70-131), marginalizeRelative (…Analytic.cpp:665-809 +
marginalization.cpp:532-588) and a pose graph over the recovered relative-pose factors, restated in oracle/."""
import numpy as np
import pytest

from oracle import twin
from sadvio_amd import capi, synthetic


def rand_T(rng, scale=1.0):
    R = synthetic.exp_so3(scale * rng.standard_normal(3))
    return np.concatenate([R.reshape(9), 2.0 * rng.standard_normal(3)])


def rel_window(Ts, factors, fixed=(0,), priors=()):
    """A pose-graph window: key-frames only (their slots hold frame-to-world poses), Relative6DPose factors."""
    n = len(Ts)
    kc = np.zeros(n, dtype=np.uint8)
    for k in fixed:
        kc[k] = 1
    eye12 = np.concatenate([np.eye(3).ravel(), np.zeros(3)])
    w = capi.FlatWindow(kf_T_f_w=np.array(Ts), kf_const=kc, cam_K=np.array([[400.0, 400.0, 320.0, 240.0]]), cam_T_s_f=eye12[None, :].copy(),
                        cam_sigma=np.ones(1),   # one (unused) camera: a pose graph has no observations
                        lmk_p=np.zeros((0, 3)), lmk_obs_ptr=np.zeros(1, dtype=np.int32), obs_kf=np.zeros(0, dtype=np.int32),
                        obs_cam=np.zeros(0, dtype=np.int32), obs_meas=np.zeros((0, 2)), factor_type=capi.FACTOR_PIXEL)
    w.sparse_priors = list(factors)
    w.pose_priors = list(priors)
    return w


def test_relative_pose_factor_against_50_digits_and_numeric_jacobian(oracle_lib):
    mp = pytest.importorskip("mpmath")
    rng = np.random.default_rng(4)
    B = twin.Backend("mp", 50)
    for trial in range(4):
        Ta, Tb = rand_T(rng), rand_T(rng)
        Tab = rand_T(rng, 0.3)
        W = np.eye(6) + 0.2 * rng.standard_normal((6, 6))
        da, db = 0.05 * rng.standard_normal(6), 0.05 * rng.standard_normal(6)
        w = rel_window([Ta, Tb], [dict(type=capi.SPARSE_RELATIVE_POSE, kf=0, kf_b=1, T_prior=Tab, sqrt_inf=W)], fixed=())
        xp = np.stack([da, db])
        r, J = oracle_lib.sparse_factor(w, 0, xp=xp)
        rm, Jam, Jbm = twin.relative_pose_factor(B, Ta, Tb, Tab, W, da, db)
        assert np.abs(r - B.f(rm)).max() < 1e-12
        assert np.abs(J[:, :6] - B.f(Jam)).max() < 1e-11 and np.abs(J[:, 6:12] - B.f(Jbm)).max() < 1e-11
    # the reference's acceptance criterion for its factors (residual_test.cpp:124): analytic vs central differences
    Ta, Tb = rand_T(rng), rand_T(rng)
    Tab = rand_T(rng, 0.3)
    W = np.eye(6)
    w = rel_window([Ta, Tb], [dict(type=capi.SPARSE_RELATIVE_POSE, kf=0, kf_b=1, T_prior=Tab, sqrt_inf=W)], fixed=())
    x0 = np.zeros((2, 6))
    r0, J0 = oracle_lib.sparse_factor(w, 0, xp=x0)
    Jn = np.zeros((6, 12))
    h = 1e-6
    for c in range(12):
        xp, xm = x0.copy(), x0.copy()
        xp[c // 6, c % 6] += h; xm[c // 6, c % 6] -= h
        Jn[:, c] = (oracle_lib.sparse_factor(w, 0, xp=xp)[0] - oracle_lib.sparse_factor(w, 0, xp=xm)[0]) / (2 * h)
    # as coded, the rotation-by-translation blocks d(log R) / d(t) are zero (true) and every coded block matches the
    # numeric derivative at zero deltas
    assert np.abs(J0[:, :12] - Jn).max() < 1e-5
    # residual zero at T_a_b_prior = T_a^-1 T_b
    Tab_exact = synthetic.T_to_12(synthetic.inv4(synthetic.T12_to_4(Ta)) @ synthetic.T12_to_4(Tb))
    w = rel_window([Ta, Tb], [dict(type=capi.SPARSE_RELATIVE_POSE, kf=0, kf_b=1, T_prior=Tab_exact, sqrt_inf=W)], fixed=())
    assert np.abs(oracle_lib.sparse_factor(w, 0, xp=x0)[0]).max() < 1e-12


def dense_relative(w, a, b, oracle_lib):
    """The reference's algorithm with a DENSE eigen-decomposition of Amm (what Eigen does), in NumPy."""
    r, Jp, Jl, _ = oracle_lib.linearize(w)
    entries, first, last = [], {}, 0
    for l in range(w.n_lmk):
        o = range(w.lmk_obs_ptr[l], w.lmk_obs_ptr[l + 1])
        if not any(w.obs_kf[q] == a for q in o):
            continue
        for q in o:
            if w.obs_kf[q] == b:
                first.setdefault(l, last)
                entries.append((l, first[l]))
                last += 3
    m = last
    A = np.zeros((m + 12, m + 12))
    for l, lc in entries:
        for q in range(w.lmk_obs_ptr[l], w.lmk_obs_ptr[l + 1]):
            if w.obs_kf[q] not in (a, b):
                continue
            J = np.zeros((2, m + 12))
            pc = m + (0 if w.obs_kf[q] == a else 6)
            J[:, pc:pc + 6] = Jp[q]; J[:, lc:lc + 3] = Jl[q]
            A += J.T @ J
    Amm = 0.5 * (A[:m, :m] + A[:m, :m].T)
    ev, V = np.linalg.eigh(Amm)
    cut = max(1e-12, m * np.finfo(float).eps * ev.max())
    inv = V @ np.diag(np.where(ev > cut, 1 / np.where(ev > cut, ev, 1), 0)) @ V.T
    return A[m:, m:] - A[m:, :m] @ inv @ A[m:, :m].T, m


def test_marginalize_relative_matches_dense_algorithm(oracle_lib):
    w = synthetic.make_window(n_kf=4, n_lmk=160, obs_per_lmk=6, seed=13, pixel_noise=0.3)
    inf, Ak, m = oracle_lib.marginalize_relative(w, 1, 2)
    Ak_np, m_np = dense_relative(w, 1, 2, oracle_lib)
    assert m == m_np and m > 60
    assert np.abs(Ak - Ak_np).max() <= 1e-9 * np.abs(Ak_np).max()
    assert np.abs(inf - inf.T).max() <= 1e-8 * np.abs(inf).max()
    assert np.linalg.eigvalsh(0.5 * (inf + inf.T)).min() > 0        # two stereo frames sharing landmarks: 6 constrained dof
    # no shared landmark: refused
    w2 = synthetic.make_window(n_kf=12, n_lmk=60, obs_per_lmk=3, seed=2, band=1, length=40.0)
    assert oracle_lib.marginalize_relative(w2, 0, 11) is None


def perturbed(Twf, rng, rot=0.02, trans=0.1, skip=(0,)):
    out = []
    for k, T in enumerate(Twf):
        d = np.zeros(6) if k in skip else np.concatenate([rot * rng.standard_normal(3), trans * rng.standard_normal(3)])
        D = np.eye(4); D[:3, :3] = synthetic.exp_so3(d[:3]); D[:3, 3] = d[3:]
        out.append(synthetic.T_to_12(synthetic.T12_to_4(T) @ D))
    return out


def compose(T12, d6):
    D = np.eye(4); D[:3, :3] = synthetic.exp_so3(d6[:3]); D[:3, 3] = d6[3:]
    return synthetic.T12_to_4(T12) @ D


def relative_prior(Ta_wf, Tb_wf):
    """T_a_b = T_a_w T_w_b from two frame-to-world poses."""
    return synthetic.T_to_12(synthetic.inv4(synthetic.T12_to_4(Ta_wf)) @ synthetic.T12_to_4(Tb_wf))


def loop_graph(n, rng):
    """Ground-truth frame-to-world poses on a loop + exact relative-pose factors (chain, skip-one links, loop closure) with
    well-conditioned random sqrt informations."""
    Twf = []
    for k in range(n):
        a = 2 * np.pi * k / n
        T = np.eye(4)
        T[:3, :3] = synthetic.exp_so3(np.array([0.1 * np.sin(a), 0.2 * np.cos(a), a]))
        T[:3, 3] = [4 * np.cos(a), 4 * np.sin(a), 0.3 * np.sin(2 * a)]
        Twf.append(synthetic.T_to_12(T))
    factors = []
    for k in range(n):
        for step in (1, 2):
            b = (k + step) % n
            W = np.diag(rng.uniform(5, 50, 6)) + rng.standard_normal((6, 6))
            factors.append(dict(type=capi.SPARSE_RELATIVE_POSE, kf=k, kf_b=b, T_prior=relative_prior(Twf[k], Twf[b]), sqrt_inf=W))
    return Twf, factors


def test_pose_graph_recovers_the_poses(oracle_lib):
    rng = np.random.default_rng(8)
    Twf, factors = loop_graph(12, rng)
    pert = perturbed(Twf, rng)
    g = rel_window(pert, factors, fixed=(0,))
    opts = capi.reference_options(); opts.max_num_iterations = 50; opts.function_tolerance = 1e-14
    res = oracle_lib.solve(g, opts)
    assert res["summary"].final_cost < 1e-14 * res["summary"].initial_cost
    for k in range(len(Twf)):
        assert np.abs(compose(pert[k], res["pose"][k]) - synthetic.T12_to_4(Twf[k])).max() < 1e-8
    # a free gauge held by a pose prior instead of a constant key-frame
    g2 = rel_window(pert, factors, fixed=(), priors=[(0, Twf[0], 100.0 * np.ones(6))])
    res2 = oracle_lib.solve(g2, opts)
    for k in range(len(Twf)):
        assert np.abs(compose(pert[k], res2["pose"][k]) - synthetic.T12_to_4(Twf[k])).max() < 1e-6


def nfr_chain(oracle_lib, w):
    """Relative6DPose factors between consecutive key-frames of a visual window, information from marginalizeRelative."""
    Twf = [synthetic.T_to_12(synthetic.inv4(synthetic.T12_to_4(T))) for T in w.kf_T_f_w]      # frame-to-world
    factors, infos = [], []
    for k in range(w.n_kf - 1):
        out = oracle_lib.marginalize_relative(w, k, k + 1)
        assert out is not None
        inf = 0.5 * (out[0] + out[0].T)
        ev, V = np.linalg.eigh(inf)
        Wm = (V * np.sqrt(np.maximum(ev, 0.0))) @ V.T            # symmetric square root
        factors.append(dict(type=capi.SPARSE_RELATIVE_POSE, kf=k, kf_b=k + 1, T_prior=relative_prior(Twf[k], Twf[k + 1]), sqrt_inf=Wm))
        infos.append(inf)
    return Twf, factors, infos


def test_pose_graph_over_recovered_factors(oracle_lib):
    """The chain the §8f row describes: consecutive key-frames of a visual window tied by Relative6DPose factors whose
    information comes from marginalizeRelative; perturbed poses return to the window's (exact-measurement factors).
    The recovered information has six healthy eigenvalues: the gauge null space of Ak is cut at the noise floor of the
    Schur complement (oracle/marg.c) instead of the reference's absolute 1e-12, which keeps rounding noise."""
    w = synthetic.make_window(n_kf=6, n_lmk=400, obs_per_lmk=6, seed=21, pixel_noise=0.0, rot_perturb_deg=0.0, trans_perturb=0.0, lmk_perturb=0.0)
    Twf, factors, infos = nfr_chain(oracle_lib, w)
    for inf in infos:
        ev = np.linalg.eigvalsh(inf)
        assert ev.min() > 1e-7 * ev.max()
    pert = perturbed(Twf, np.random.default_rng(5))
    g = rel_window(pert, factors, fixed=(0,))
    opts = capi.reference_options(); opts.max_num_iterations = 50; opts.function_tolerance = 1e-14
    res = oracle_lib.solve(g, opts)
    assert res["summary"].initial_cost > 1e3 and res["summary"].final_cost < 1e-12 * res["summary"].initial_cost
    for k in range(len(Twf)):
        assert np.abs(compose(pert[k], res["pose"][k]) - synthetic.T12_to_4(Twf[k])).max() < 1e-6
    # the information does not depend on rounding noise: the same frames seen through a window whose landmark order is
    # reversed (different summation order in every accumulation) give the same matrix
    rev = synthetic.make_window(n_kf=6, n_lmk=400, obs_per_lmk=6, seed=21, pixel_noise=0.0, rot_perturb_deg=0.0, trans_perturb=0.0, lmk_perturb=0.0)
    order = np.arange(rev.n_lmk)[::-1]
    ptr = [0]; okf = []; ocam = []; meas = []
    for l in order:
        o = slice(rev.lmk_obs_ptr[l], rev.lmk_obs_ptr[l + 1])
        okf += list(rev.obs_kf[o]); ocam += list(rev.obs_cam[o]); meas += list(rev.obs_meas[o]); ptr.append(len(okf))
    rev.lmk_p = rev.lmk_p[order].copy(); rev.lmk_obs_ptr = np.array(ptr, dtype=np.int32); rev.obs_kf = np.array(okf, dtype=np.int32)
    rev.obs_cam = np.array(ocam, dtype=np.int32); rev.obs_meas = np.array(meas); rev.lmk_id = None
    for k in (0, 3):
        a = oracle_lib.marginalize_relative(w, k, k + 1)[0]
        b = oracle_lib.marginalize_relative(rev, k, k + 1)[0]
        assert np.abs(a - b).max() <= 1e-6 * np.abs(a).max()
