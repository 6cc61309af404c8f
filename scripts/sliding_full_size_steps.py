"""Per-step table of the device's config-3-size dense sliding sequence against the oracle fixture (tests/golden/sliding_config3_size_dense.npz):
rank, iterations, termination, final cost of both sides, pose-delta and trajectory differences. DEV_FORM=eigen runs the device with the eigen form of the prior."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sadvio_amd import capi
from test_gpu_sliding_long import run_sequence
SEED = int(os.environ.get("SEED", "977"))
gold = np.load(os.path.join(ROOT, "tests", "golden", "sliding_config3_size_dense%s.npz" % ("" if SEED == 977 else "_s%d" % SEED)))
n_win, n_kf, n_lmk, keep_cap, n_steps = (int(v) for v in gold["params"])
rec = []
def snap(step, side, st, kfs2, result, rank):
    it, term, cost, d = result
    k = len(rec)
    rec.append(1)
    print(f"step {step}: rank {rank} gold {tuple(gold['rank'][k])} it {it}/{gold['it'][k]} term {term}/{gold['term'][k]} cost dev {cost:.6f} gold {gold['cost'][k]:.6f} "
          f"dpose {np.abs(np.array(d['pose']) - gold['pose'][k]).max():.2e} drift {np.abs(st['T'] - gold['T'][k]).max():.2e}", flush=True)
run_sequence(capi.Backend, None, True, False, "reference", n_steps=n_steps, run=("dev",), snap=snap, n_win=n_win, n_kf=n_kf, n_lmk=n_lmk, length=float(gold["length"]), keep_cap=keep_cap, dev_form=os.environ.get("DEV_FORM", "cholesky"), seed=SEED)
