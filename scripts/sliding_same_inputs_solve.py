"""Step STEP of the config-3-size dense sequence on the device; then the ORACLE's solve of the SAME window with the device's own prior (read
back): solve parity on identical inputs, at the step where the two separately propagated sequences part."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sadvio_amd import capi
from oracle import oracle
import test_gpu_sliding_long as T
gold = np.load(os.path.join(ROOT, "tests", "golden", "sliding_config3_size_dense.npz"))
n_win, n_kf, n_lmk, keep_cap, n_steps = (int(v) for v in gold["params"])
STEP = int(sys.argv[1]) if len(sys.argv) > 1 else 13
cap = {}
class BE(capi.Backend):
    def set_windows(self, ws):
        self.last_ws = ws
        return super().set_windows(ws)
    def solve(self, opts):
        if len(cap) == 0 and getattr(self, "n_solves", 0) == STEP:
            cap["prior"] = self.get_prior()
        r = super().solve(opts)
        self.n_solves = getattr(self, "n_solves", 0) + 1
        self.last_trace = self.get_trace(0)
        return r
holder = {}
def mk(device=0):
    holder["be"] = BE(device=device)
    return holder["be"]
def snap(step, side, st, kfs2, result, rank):
    if step == STEP:
        cap["w2"] = holder["be"].last_ws[0]; cap["res"] = result; cap["trace"] = holder["be"].last_trace.copy()
T.run_sequence(mk, None, True, False, "reference", n_steps=STEP + 1, run=("dev",), snap=snap, n_win=n_win, n_kf=n_kf, n_lmk=n_lmk, length=float(gold["length"]), keep_cap=keep_cap)
w2 = cap["w2"]; pr = cap["prior"]
print("device prior", pr["n_full"], pr["n"], pr["form"])
dp = dict(w2.dense_prior, J=pr["J"], r0=pr["r0"])
t = time.time()
r = oracle.solve(w2, capi.reference_options(), dense_prior=dp)
it_d, term_d, cost_d, d = cap["res"]
print(f"oracle solve of the device's window + prior: {time.time() - t:.0f} s; it {r['summary'].iterations}/{it_d} term {r['summary'].termination}/{term_d} cost oracle {r['summary'].final_cost:.9f} device {cost_d:.9f} "
      f"|dpose| {np.abs(r['pose'] - d['pose']).max():.2e} |dlmk| {np.abs(r['lmk'] - d['lmk']).max():.2e}; fixture cost {gold['cost'][STEP]:.9f}")
np.set_printoptions(linewidth=250, precision=12)
lo, tr = r["log"], cap["trace"]
n = min(len(lo), len(tr))
print("per iteration: cost device | cost oracle | accepted device/oracle | radius device | radius oracle")
for k in range(n):
    print(f"  {k:2d} {tr[k,0]:.9f} {lo[k,0]:.9f}  {int(tr[k,5])}/{int(lo[k,5])}  {tr[k,6]:.6e} {lo[k,6]:.6e}  cost_change {tr[k,1]:.6e} {lo[k,1]:.6e}")
