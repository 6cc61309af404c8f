"""How far the DEVICE's own config-3-size dense sequence moves when every measurement of the trajectory is nudged by one unit in the last
place (three seeded draws against the un-nudged run, all on the device): the amplification of each step, without any second implementation.
A step whose solve runs into the iteration cap or along a validity boundary multiplies a 1e-11 difference by 1e3 and more; the
device-against-oracle difference of tests/test_gpu_sliding_full_size.py has to be read against this. SEED selects the trajectory."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sadvio_amd import capi
from test_gpu_sliding_long import run_sequence
SEED = int(os.environ.get("SEED", "977"))
gold = np.load(os.path.join(ROOT, "tests", "golden", "sliding_config3_size_dense%s.npz" % ("" if SEED == 977 else "_s%d" % SEED)))
n_win, n_kf, n_lmk, keep_cap, n_steps = (int(v) for v in gold["params"])
runs = []
for nudge in (None, 1, 2, 3):
    rec = []
    def snap(step, side, st, kfs2, result, rank):
        it, term, cost, d = result
        rec.append(dict(it=it, term=term, cost=cost, rank=rank, pose=np.array(d["pose"]).copy(), T=st["T"].copy()))
    run_sequence(capi.Backend, None, True, False, "reference", n_steps=n_steps, run=("dev",), snap=snap, n_win=n_win, n_kf=n_kf, n_lmk=n_lmk, length=float(gold["length"]), keep_cap=keep_cap, nudge_seed=nudge, seed=SEED)
    runs.append(rec)
base = runs[0]
for k in range(n_steps):
    dp = max(float(np.abs(r[k]["pose"] - base[k]["pose"]).max()) for r in runs[1:])
    dT = max(float(np.abs(r[k]["T"] - base[k]["T"]).max()) for r in runs[1:])
    dc = max(abs(r[k]["cost"] - base[k]["cost"]) / base[k]["cost"] for r in runs[1:])
    same = all((r[k]["it"], r[k]["term"], tuple(r[k]["rank"])) == (base[k]["it"], base[k]["term"], tuple(base[k]["rank"])) for r in runs[1:])
    print(f"step {k}: it {base[k]['it']} term {base[k]['term']} rank {base[k]['rank']}: nudged runs differ by |dpose| {dp:.1e}, trajectory {dT:.1e}, cost {dc:.1e}; same counts: {same}", flush=True)
