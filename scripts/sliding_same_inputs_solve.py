"""The config-3-size dense sliding sequence on the device; at every requested step the ORACLE then solves the SAME window with the device's
own prior (read back): solve parity on IDENTICAL inputs at the shipped window size (12 KF, ~ 3 000 landmarks in view, a dense prior of
915 .. 972 columns; N_p ~ 1 100), attempt by attempt. Usage: python scripts/sliding_same_inputs_solve.py [first_step] [last_step] [--log]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sadvio_amd import capi
from oracle import oracle
import test_gpu_sliding_long as T
from golden_util import lmk_err
gold = np.load(os.path.join(ROOT, "tests", "golden", "sliding_config3_size_dense.npz"))
n_win, n_kf, n_lmk, keep_cap, n_steps = (int(v) for v in gold["params"])
VIO = os.environ.get("VIO", "1") != "0"       # VIO=0 N_WIN=20 N_KF=34 N_LMK=12000: a VO sequence of config-2 shaped windows
n_win, n_kf, n_lmk = int(os.environ.get("N_WIN", n_win)), int(os.environ.get("N_KF", n_kf)), int(os.environ.get("N_LMK", n_lmk))
LENGTH = float(os.environ.get("LENGTH", float(gold["length"])))
SEED = int(os.environ.get("SEED", "977"))     # the trajectory
args_ = [a for a in sys.argv[1:] if not a.startswith("--")]
S0 = int(args_[0]) if len(args_) > 0 else 13
S1 = int(args_[1]) if len(args_) > 1 else S0
cap = {}
class BE(capi.Backend):
    def set_windows(self, ws):
        self.last_ws = ws
        return super().set_windows(ws)
    def solve(self, opts):
        k = getattr(self, "n_solves", 0)
        if S0 <= k <= S1:
            cap[k] = dict(prior=self.get_prior())
        r = super().solve(opts)
        self.n_solves = k + 1
        self.last_trace = self.get_trace(0)
        return r
holder = {}
def mk(device=0):
    holder["be"] = BE(device=device)
    return holder["be"]
def snap(step, side, st, kfs2, result, rank):
    if step in cap:
        cap[step].update(w2=holder["be"].last_ws[0], res=result, trace=holder["be"].last_trace.copy())
T.run_sequence(mk, None, VIO, False, "reference", n_steps=S1 + 1, run=("dev",), snap=snap, n_win=n_win, n_kf=n_kf, n_lmk=n_lmk, length=LENGTH, keep_cap=keep_cap, seed=SEED)
worst = 0.0
for step in range(S0, S1 + 1):
    c = cap[step]
    w2, pr = c["w2"], c["prior"]
    dp = dict(w2.dense_prior, J=pr["J"], r0=pr["r0"])
    t = time.time()
    r = oracle.solve(w2, capi.reference_options(), dense_prior=dp)
    it_d, term_d, cost_d, d = c["res"]
    lo, tr = r["log"], c["trace"]
    n = min(len(lo), len(tr))
    same_attempts = bool(np.array_equal(tr[:n, 5], lo[:n, 5])) and len(lo) == len(tr)
    dpose = float(np.abs(r["pose"] - d["pose"]).max())
    worst = max(worst, dpose)
    print(f"step {step}: prior {pr['n_full']} of {pr['n']}, {w2.n_lmk} landmarks, {w2.n_obs} observations; iterations {r['summary'].iterations}/{it_d} termination {r['summary'].termination}/{term_d} "
          f"accept / reject pattern equal: {same_attempts} ({int((lo[1:n - 1, 5] == 0).sum())} rejected attempts); |dpose| {dpose:.2e}, cost relative {abs(r['summary'].final_cost - cost_d) / cost_d:.1e}, "
          f"landmarks (relative beyond a metre) {lmk_err(d['lmk'], r['lmk']):.1e} ({time.time() - t:.0f} s)", flush=True)
    if "--log" in sys.argv:
        for k in range(n):
            print(f"  {k:2d} {tr[k,0]:.9f} {lo[k,0]:.9f}  {int(tr[k,5])}/{int(lo[k,5])}  {tr[k,6]:.6e} {lo[k,6]:.6e}  cost_change {tr[k,1]:.6e} {lo[k,1]:.6e}")
print(f"worst pose difference over steps {S0} .. {S1}: {worst:.2e}")
