"""Step STEP of the config-3-size dense sequence on the device; the ORACLE then marginalises the SAME window with the device's previous
prior (read back) as `last`: marginalisation parity on identical inputs at the step where the separately propagated sequences part."""
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
    def marginalize(self, *a, **k):
        self.n_marg = getattr(self, "n_marg", 0)
        if self.n_marg == STEP:
            cap["prev"] = self.get_prior()
        g = super().marginalize(*a, **k)
        if self.n_marg == STEP:
            cap["new"] = self.get_prior()
        self.n_marg += 1
        return g
def hook(step, side, w, g, args):
    if step == STEP:
        cap["w"] = w; cap["args"] = args; cap["g"] = g
T.run_sequence(BE, None, True, False, "reference", n_steps=STEP + 1, run=("dev",), hook=hook, n_win=n_win, n_kf=n_kf, n_lmk=n_lmk, length=float(gold["length"]), keep_cap=keep_cap)
prev, new, args = cap["prev"], cap["new"], dict(cap["args"])
print("device: previous prior", prev["n_full"], prev["n"], "new prior", new["n_full"], new["n"])
args["last"] = dict(args["last"], J=prev["J"], r0=prev["r0"])
t = time.time()
go = oracle.marginalize(cap["w"], **args)
print(f"oracle marginalize of the device's window + previous prior: {time.time() - t:.0f} s, rank {go['n_full']} of {go['n']}")
Ad, Ao = new["J"].T @ new["J"], go["J"].T @ go["J"]
bd, bo = new["J"].T @ new["r0"], go["J"].T @ go["r0"]
print(f"|J^T J dev - ora| / max {np.abs(Ad - Ao).max() / np.abs(Ao).max():.2e} (absolute {np.abs(Ad - Ao).max():.2e}); |J^T r0 dev - ora| / max {np.abs(bd - bo).max() / np.abs(bo).max():.2e} (absolute {np.abs(bd - bo).max():.2e}); "
      f"|r0|^2 dev {new['r0'] @ new['r0']:.9f} ora {go['r0'] @ go['r0']:.9f}")
ev = np.linalg.eigvalsh(0.5 * (Ao + Ao.T))
print("smallest eigenvalues of the oracle's information (LAPACK):", ev[:6])
D = Ad - Ao
evd = np.linalg.eigvalsh(0.5 * (D + D.T))
print("eigenvalues of the difference dev - ora: min", evd[0], "max", evd[-1])
w_, V = np.linalg.eigh(0.5 * (D + D.T))
v = V[:, 0]
top = np.argsort(-np.abs(v))[:8]
print("direction of the largest difference: columns", [(int(c), round(float(v[c]), 4)) for c in top])
last = args["last"]
print("previous prior columns of the kept frame:", last.get("kf_col"), "; previous-prior landmarks that LEFT the window (col -1):", int((np.asarray(last["lmk_col"]) < 0).sum()), "of", len(last["lmk_col"]))
print("new prior: kf_col", cap["g"]["kf_col"], "lmk_col head", list(cap["g"]["lmk_col"])[:6])
