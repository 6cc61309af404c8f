"""The config-3-size dense sliding sequence on the device; at every requested step the ORACLE then marginalises the SAME window with the
device's previous prior (read back) as `last`: marginalisation parity on IDENTICAL inputs at the shipped window size, step by step — nothing
is inherited from separately propagated states. This comparison located the Amm pseudo-inverse defect of round 6 (step 13: 3.7e-5).
Usage: python scripts/sliding_same_inputs_marg.py [first_step] [last_step]   (GPU box; ~ 20 - 40 s of oracle per step)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sadvio_amd import capi
from oracle import oracle
import test_gpu_sliding_long as T
gold = np.load(os.path.join(ROOT, "tests", "golden", "sliding_config3_size_dense.npz"))
n_win, n_kf, n_lmk, keep_cap, n_steps = (int(v) for v in gold["params"])
VIO = os.environ.get("VIO", "1") != "0"       # VIO=0 N_WIN=20 N_KF=34 N_LMK=12000: a VO sequence of config-2 shaped windows
n_win, n_kf, n_lmk = int(os.environ.get("N_WIN", n_win)), int(os.environ.get("N_KF", n_kf)), int(os.environ.get("N_LMK", n_lmk))
LENGTH = float(os.environ.get("LENGTH", float(gold["length"])))
SEED = int(os.environ.get("SEED", "977"))     # the trajectory
S0 = int(sys.argv[1]) if len(sys.argv) > 1 else 13
S1 = int(sys.argv[2]) if len(sys.argv) > 2 else S0
cap = {}
class BE(capi.Backend):
    def marginalize(self, *a, **k):
        self.n_marg = getattr(self, "n_marg", 0)
        want = S0 <= self.n_marg <= S1
        prev = self.get_prior() if want else None
        g = super().marginalize(*a, **k)
        if want:
            cap[self.n_marg] = dict(prev=prev, new=self.get_prior())
        self.n_marg += 1
        return g
def hook(step, side, w, g, args):
    if step in cap:
        cap[step].update(w=w, args=args, g=g)
T.run_sequence(BE, None, VIO, False, "reference", n_steps=S1 + 1, run=("dev",), hook=hook, n_win=n_win, n_kf=n_kf, n_lmk=n_lmk, length=LENGTH, keep_cap=keep_cap, seed=SEED, dev_form=os.environ.get("DEV_FORM", "cholesky"))
worst = 0.0
for step in range(S0, S1 + 1):
    c = cap[step]
    prev, new, args = c["prev"], c["new"], dict(c["args"])
    if args.get("last") is not None:
        args["last"] = dict(args["last"], J=prev["J"], r0=prev["r0"])
    t = time.time()
    go = oracle.marginalize(c["w"], **args)
    Ad, Ao = new["J"].T @ new["J"], go["J"].T @ go["J"]
    bd, bo = new["J"].T @ new["r0"], go["J"].T @ go["r0"]
    eH = np.abs(Ad - Ao).max() / np.abs(Ao).max(); eg = np.abs(bd - bo).max() / np.abs(bo).max()
    worst = max(worst, eH)
    print(f"step {step}: previous prior {prev['n_full'] if prev['valid'] else 0} of {prev['n'] if prev['valid'] else 0}; device {new['n_full']} of {new['n']}, oracle {go['n_full']} of {go['n']} ({time.time() - t:.0f} s): "
          f"|J^T J dev - ora| / max {eH:.2e}, |J^T r0 dev - ora| / max {eg:.2e}, |r0|^2 {new['r0'] @ new['r0']:.6f} / {go['r0'] @ go['r0']:.6f}", flush=True)
print(f"worst information difference over steps {S0} .. {S1}: {worst:.2e}")
