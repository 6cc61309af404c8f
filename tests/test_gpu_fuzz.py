"""Seeded randomised parity sweep of the HIP path against the oracle (the pinned part of scripts/gpu_fuzz.py): a fixed
list of random submissions — VO / VIO, pixel / bearing factors, constant masks, pose / dense / sparse priors, Huber on or
off, batches of 1-3, with and without hipGraph — plus the windows earlier sweeps disagreed on.

Tolerances. Iteration counts, terminations: identical. Pose deltas: 1e-6 (BASELINE.json) — on every direction the data
determine; a direction only the LM damping holds (relative stiffness < 1e-9 in the damped reduced system, e.g. the 4-D null
space of a key-frame with one observation) gets the rounding-amplification allowance of tests/conditioning.py, and a window
whose LM trajectory amplifies a 1-ulp nudge of its own measurements beyond the fixed bars (long valleys at radius 1e11) is held
to 8x the oracle's own sensitivity to that nudge instead (oracle_self_sensitivity); both are consulted
only when the strict bar fails (round 2's one pose disagreement, seed 39573273, pinned below and arbitrated against the
long-double twin in test_pose_seed_against_long_double_twin). Landmark deltas: LMK_TOL = 1e-5,
RELATIVE to the landmark's own delta once that exceeds 1 m: the sweeps' only landmark disagreements (up to 1.6e-3 m)
were all on landmarks the optimisation itself sends away — 3-view tracks with (numerically) collinear rays whose H_ll has
an eigenvalue of 1e-18..1e-21 and whose delta grows to 2e1 .. 8e5 m while poses and cost agree to 1e-13: an absolute
1e-5 on a 7.8e5 m delta would be 1e-11 relative, beyond double precision times that conditioning, for either side
(gpurun_out/fuzz_r02a.log). Costs: 1e-8 relative."""
import numpy as np
import pytest

import conditioning
import fuzz_helpers as fz
from sadvio_amd import capi

pytestmark = pytest.mark.gpu
POSE_TOL, LMK_TOL, COST_RTOL = 1e-6, 1e-5, 1e-8
# caps of the rounding-sensitivity allowance (only consulted after the fixed bars fail, and only while the 1-ulp-nudged oracle
# still takes the same LM path): the three windows of ~30 000 swept that need it measure 1.7e-5 / 2.9e-8 at worst (seeds
# 234496164, 319802335, 261931991) — anything beyond these caps fails and has to be arbitrated explicitly.
ALLOW_POSE, ALLOW_COST, ALLOW_LMK = 1e-4, 1e-6, 1e-3
ALLOWANCE_LOG = []   # every use of the allowance in this session (printed, and summarised by test_zz_allowance_report)

# the windows randomised sweeps disagreed on live in tests/golden/fuzz_pinned.json (specs = generator parameters only):
# runaway landmarks; dense priors whose rejected steps exposed an oracle bug (fixed since)


def check_case(case, oracle_lib):
    ws = [fz.build_window(s) for s in case["specs"]]
    opts = fz.options(case)
    be = capi.Backend(device=0, use_graph=case["use_graph"])
    try:
        be.set_windows(ws)
        sums = be.solve(opts)
        for k, w in enumerate(ws):
            d = be.get_deltas(k)
            ref = oracle_lib.solve(w, opts, dense_prior=w.dense_prior)
            rs = ref["summary"]
            what = fz.describe(case["specs"][k])
            assert (sums[k].iterations, sums[k].termination) == (rs.iterations, rs.termination), what
            dp = float(np.abs(d["pose"] - ref["pose"]).max())
            dc = abs(sums[k].final_cost - rs.final_cost) / abs(rs.final_cost)
            dl = 0.0
            if w.n_lmk:
                scale = np.maximum(1.0, np.abs(ref["lmk"]).max(axis=1))
                dl = float((np.abs(d["lmk"] - ref["lmk"]).max(axis=1) / scale).max())
            if dp > POSE_TOL or dc > COST_RTOL or dl > LMK_TOL:
                # the fixed bars failed: is the WINDOW that sensitive? (i) the oracle against itself under a 1-ulp nudge of the
                # measurements, (ii) for the poses alone, the damping-held directions of the reduced system. The allowance is
                # CAPPED: it only exists while the nudged oracle itself still takes the same LM path (`same`), and it
                # never exceeds ALLOW_* — a window beyond that has to be pinned and arbitrated against the long-double twin.
                sp, sc, sl, same = conditioning.oracle_self_sensitivity(lambda s_=case["specs"][k]: fz.build_window(s_), opts, oracle_lib, ref)
                a_pose = min(conditioning.SELF_K * sp, ALLOW_POSE) if same else 0.0
                a_cost = min(conditioning.SELF_K * sc, ALLOW_COST) if same else 0.0
                a_lmk = min(conditioning.SELF_K * sl, ALLOW_LMK) if same else 0.0
                cond_ok, report = (False, "")
                if dp > max(POSE_TOL, a_pose):
                    cond_ok, report = conditioning.pose_difference_within_conditioning(w, oracle_lib, ref, d["pose"], ref["pose"], POSE_TOL)
                    cond_ok = cond_ok and dc <= 1e-8   # conditioning.py: the failing window must also agree on the cost to 1e-8
                ALLOWANCE_LOG.append({"window": what, "dpose": dp, "dcost": dc, "dlmk": dl, "oracle_self": [sp, sc, sl], "same_path": bool(same),
                                      "by": "eigen-direction bound" if cond_ok else "oracle self-sensitivity"})
                print(f"[fuzz] conditioning allowance used: {ALLOWANCE_LOG[-1]}")
                assert dp <= max(POSE_TOL, a_pose) or cond_ok, (what, dp, sp, same, report)
                assert dc <= max(COST_RTOL, a_cost), (what, dc, sc, same)
                assert dl <= max(LMK_TOL, a_lmk), (what, dl, sl, same)
            if w.has_imu:
                for key in ("dv", "dba", "dbg"):
                    assert np.abs(d[key] - ref[key]).max() <= POSE_TOL, (what, key)
    finally:
        be.close()


@pytest.mark.parametrize("seed", [12345, 777, 20250404])
def test_random_submissions(oracle_lib, seed):
    rng = np.random.default_rng(seed)
    for _ in range(40):
        check_case(fz.draw_case(rng), oracle_lib)


def _pinned():
    import json, os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_pinned.json")))


@pytest.mark.parametrize("lm", [0, 1])
def test_pinned_disagreements(oracle_lib, monkeypatch, lm):
    """Every window a sweep ever flagged (generator parameters only), on the kernels it was flagged on — and all of them
    again with the throughput kernels forced (SADVIO_LM=1; windows with features that path does not take fall back)."""
    if lm:
        monkeypatch.setenv("SADVIO_LM", "1")
    for b in _pinned():
        if b.get("arbiter"):
            continue   # beyond the caps of the allowance: held against its long-double arbiter in test_chaotic_window_against_long_double_twin
        if lm or not b.get("lm"):
            check_case(dict(specs=[b["spec"]], huber=b["huber"], use_graph=b["use_graph"]), oracle_lib)


def test_flagged_landmarks_are_runaway(oracle_lib):
    """The sweeps' landmark disagreements (1e-5 .. 0.17 m absolute) must all sit on landmarks the optimisation itself sends
    away (|delta| >= 1 m, where LMK_TOL is relative): any landmark with a sub-metre delta has to meet 1e-5 absolutely."""
    for b in _pinned():
        seen = b.get("seen", {})
        if seen.get("dlmk", 0.0) <= LMK_TOL:
            continue
        if seen.get("dpose", 0.0) > 1e-8 or seen.get("dcost", 0.0) > 1e-9:
            continue   # a window whose POSES are rounding-sensitive (held to the oracle's self-sensitivity above): its landmarks follow them
        w = fz.build_window(b["spec"])
        opts = fz.options(b)
        be = capi.Backend(device=0, use_graph=b["use_graph"])
        try:
            be.set_windows([w])
            be.solve(opts)
            d = be.get_deltas(0)
        finally:
            be.close()
        ref = oracle_lib.solve(w, opts, dense_prior=w.dense_prior)
        err = np.abs(d["lmk"] - ref["lmk"]).max(axis=1)
        mag = np.abs(ref["lmk"]).max(axis=1)
        sub = mag < 1.0
        assert err[sub].max(initial=0.0) <= LMK_TOL, (fz.describe(b["spec"]), float(err[sub].max()))
        assert (err[~sub] / mag[~sub]).max(initial=0.0) <= LMK_TOL, fz.describe(b["spec"])


def test_pose_seed_against_long_double_twin(oracle_lib):
    """Seed 39573273 (19 key-frames, 309 three-view landmarks; key-frame 0 carries ONE observation): the sweep's only pose
    disagreement above 1e-6. Arbiter: oracle/twin.py in long double on the un-reduced normal equations
    (tests/golden/fuzz_seed39573273_ld.npz, generated by scripts/fuzz_arbitrate.py). Every key-frame except 0 must meet the
    strict bar against the arbiter, on the device AND in the oracle; key-frame 0 may differ only inside its damping-held
    null space (tests/conditioning.py), and the device must not be further from the arbiter than 4x the oracle is."""
    import os
    b = [b for b in _pinned() if b["spec"]["seed"] == 39573273][0]
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_seed39573273_ld.npz"))
    w = fz.build_window(b["spec"])
    assert np.bincount(w.obs_kf, minlength=w.n_kf)[0] == 1
    opts = fz.options(b)
    ref = oracle_lib.solve(w, opts)
    for graph in (True, False):
        be = capi.Backend(device=0, use_graph=graph)
        try:
            be.set_windows([w])
            s = be.solve(opts)[0]
            d = be.get_deltas(0)
        finally:
            be.close()
        assert (s.iterations, s.termination) == (ref["summary"].iterations, ref["summary"].termination)
        e_dev, e_ora = np.abs(d["pose"] - z["pose"]).max(axis=1), np.abs(ref["pose"] - z["pose"]).max(axis=1)
        assert e_dev[1:].max() <= 1e-9 and e_ora[1:].max() <= 1e-9, (e_dev[1:].max(), e_ora[1:].max())
        assert np.abs(d["lmk"] - z["lmk"]).max() <= 1e-8
        assert e_dev[0] <= max(4 * e_ora[0], POSE_TOL), (e_dev[0], e_ora[0])
        for a in (d["pose"], ref["pose"]):
            ok, report = conditioning.pose_difference_within_conditioning(w, oracle_lib, ref, a, z["pose"], POSE_TOL)
            assert ok, report


def test_chaotic_window_against_long_double_twin(oracle_lib):
    """Seed 961174670 (round-4 sweep; 18 key-frames, 945 four-view landmarks, a dense prior, Huber): 20 LM iterations without
    convergence along an ill-conditioned valley. Arbiter: oracle/twin.py in LONG DOUBLE (tests/golden/fuzz_seed961174670_ld.npz,
    scripts/fuzz_arbitrate.py twin 961174670 ld schur). Round 4 found the two float64 implementations that eliminate the landmarks
    through the ADJUGATE 3 x 3 inverse and an explicitly formed (E M^-1) E^T — oracle and device — 2.1e-4 / 1.3e-5 .. 3.7e-4 (run to
    run) from the arbiter while the un-reduced float64 twin lands 3.8e-8. Round 5 (scripts/elim_numerics.py): the
    elimination in CHOLESKY form (M = L L^T, W = E L^-T, S -= W W^T: the block step of the landmark-first Cholesky CHOLMOD runs
    for the reference) closes the gap — twin 3.75e-8, oracle 3.8e-8, device 3.8e-8, every run. The bar is now ABSOLUTE:
    north_star's 1e-6 against the arbiter and against the oracle, on both launch modes."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_seed961174670_ld.npz")
    assert os.path.exists(path), "the long-double arbiter fixture is checked in: tests/golden/fuzz_seed961174670_ld.npz"
    b = [b for b in _pinned() if b["spec"]["seed"] == 961174670][0]
    w = fz.build_window(b["spec"])
    opts = fz.options(b)
    ref = oracle_lib.solve(w, opts, dense_prior=w.dense_prior)
    z = np.load(path)
    e_ora = float(np.abs(ref["pose"] - z["pose"]).max())
    e_t64 = float(np.abs(z["pose_f64_twin"] - z["pose"]).max())
    assert e_ora <= POSE_TOL, e_ora
    for graph in (True, False):
        be = capi.Backend(device=0, use_graph=graph)
        try:
            be.set_windows([w])
            s = be.solve(opts)[0]
            d = be.get_deltas(0)
        finally:
            be.close()
        assert (s.iterations, s.termination) == (ref["summary"].iterations, ref["summary"].termination)
        e_dev = float(np.abs(d["pose"] - z["pose"]).max())
        print(f"[fuzz arbiter 961174670] |pose - long double|: device {e_dev:.2e}, oracle {e_ora:.2e}, float64 twin {e_t64:.2e}")
        assert e_dev <= POSE_TOL, (e_dev, e_ora, e_t64)
        assert float(np.abs(d["pose"] - ref["pose"]).max()) <= POSE_TOL
        assert abs(s.final_cost - ref["summary"].final_cost) <= 1e-8 * ref["summary"].final_cost


def test_zz_allowance_report():
    """Keeps the list of windows that needed the conditioning allowance visible: runs last in this module, prints them,
    and fails if their number grows beyond the handful that were arbitrated (a kernel regression would show up here first)."""
    for e in ALLOWANCE_LOG:
        print("[fuzz allowance]", e)
    print(f"[fuzz allowance] {len(ALLOWANCE_LOG)} comparisons used the conditioning allowance")
    distinct = {e["window"] for e in ALLOWANCE_LOG}
    assert len(distinct) <= 6, sorted(distinct)
