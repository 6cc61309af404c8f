"""The sliding back-end loop at the SHIPPED window size (VERDICT r05 missing 6): 12-key-frame VIO windows, ~ 600 landmarks per key-frame,
300 landmarks kept in the prior per step (n = 915 .. 970 columns; /root/reference/ros/config/config.yaml:34,108), 25 key-frame steps of
marginalize -> [sparsify] -> solve -> write-back (slamBiMonoVIO.cpp:561-614) on the device, compared STEP BY STEP with the oracle's own run
of the same sequence. The oracle needs two minutes per step at this size, so its side is a committed fixture
(tests/golden/sliding_config3_size_*.npz, written by scripts/gen_sliding_golden.py from oracle/ on the CPU); the device side propagates its
own state, as in tests/test_gpu_sliding_long.py, so a disagreement compounds over the 25 steps. Bars (those of the reduced-size sequences): the
same prior rank, iteration count and termination at every step, cost to 1e-7 relative, pose deltas and the trajectories to 1e-6, landmarks
to 1e-6 relative. Round 6 found a defect with this test: the pseudo-inverse of the marginalised block was taken as a Cholesky inverse where
one eigenvalue lay below the reference's cut under pivots that all passed (step 13 of the dense sequence, behind a rank-deficient previous
prior: 1.1e-6 in the poses, 7.8e-6 by step 24; 3e-9 / 1.2e-8 with the eigenvalue bound sadvio_ba_marginalize now takes from the trace of
the inverse). Fixtures: the trajectory of the reduced-size sequences (seed 977) and a second one (979), each sparsified and dense. (Of two more
trajectories tried, one degenerates on the oracle's side itself at step 9 and one runs a solve into the iteration cap at step 22, which
multiplies the 7e-9 the sides carry into it by 760 - the device alone shows the same factor under 1-ulp nudges: profiles/r06_sliding_full_size.txt;
neither is a fixture.)"""
import os

import numpy as np
import pytest

from test_gpu_sliding_long import run_sequence

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


FIXTURES = sorted(f for f in os.listdir(GOLD) if f.startswith("sliding_config3_size_") and f.endswith(".npz"))   # one per (variant, trajectory seed)


@pytest.mark.parametrize("fixture", FIXTURES)
def test_config3_size_sequence_against_the_oracle_fixture(backend_cls, fixture):
    sparsif = "sparsified" in fixture
    gold = np.load(os.path.join(GOLD, fixture))
    n_win, n_kf, n_lmk, keep_cap, n_steps = (int(v) for v in gold["params"])
    seed = int(gold["seed"]) if "seed" in gold.files else 977
    rec = []

    def snap(step, side, st, kfs2, result, rank):
        it, term, cost, d = result
        rec.append(dict(it=it, term=term, cost=cost, rank=rank, pose=np.array(d["pose"]).copy(), T=st["T"].copy(), kfs=np.array(kfs2)))

    log, stats, sides = run_sequence(backend_cls, None, True, sparsif, "reference", n_steps=n_steps, run=("dev",), snap=snap,
                                     n_win=n_win, n_kf=n_kf, n_lmk=n_lmk, length=float(gold["length"]), keep_cap=keep_cap, seed=seed)
    assert len(rec) == n_steps == len(gold["it"])
    worst = dict(pose=0.0, cost=0.0, drift=0.0)
    for k, r in enumerate(rec):
        assert np.array_equal(r["kfs"], gold["kfs"][k])
        assert tuple(r["rank"]) == tuple(int(v) for v in gold["rank"][k]), (k, r["rank"], gold["rank"][k])
        assert (r["it"], r["term"]) == (int(gold["it"][k]), int(gold["term"][k])), (k, r["it"], r["term"], gold["it"][k], gold["term"][k])
        worst["cost"] = max(worst["cost"], abs(r["cost"] - gold["cost"][k]) / gold["cost"][k])
        worst["pose"] = max(worst["pose"], float(np.abs(r["pose"] - gold["pose"][k]).max()))
        worst["drift"] = max(worst["drift"], float(np.abs(r["T"] - gold["T"][k]).max()))
    mag = np.maximum(1.0, np.abs(gold["p"]).max(axis=1))
    rel = np.abs(sides["dev"]["p"] - gold["p"]).max(axis=1) / mag
    far = mag > 1e3
    print(f"[sliding config-3 size, {'sparsified' if sparsif else 'dense'}, trajectory {seed}] {n_steps} steps, prior columns {sorted(set(int(v[1]) for v in gold['rank']))[:1]}..{max(int(v[1]) for v in gold['rank'])}: "
          f"worst per-step |dpose| {worst['pose']:.2e}, cost {worst['cost']:.2e}, trajectory difference {worst['drift']:.2e}; landmarks within 1 km {rel[~far].max():.2e}, "
          f"beyond ({int(far.sum())}) {rel[far].max() if far.any() else 0.0:.2e}; device routes {stats}")
    assert worst["cost"] <= 1e-7 and worst["pose"] <= 1e-6 and worst["drift"] <= 1e-6, worst
    assert rel[~far].max() <= 1e-6
    assert not far.any() or rel[far].max() <= 1e-3
