#!/usr/bin/env python
"""Golden fixture of the ORACLE side of a config-3-size sliding sequence (tests/golden/sliding_config3_size.npz): 12-key-frame VIO windows,
~ 600 landmarks per key-frame, 300 landmarks kept in the prior per step (config.yaml:34,108), the reference's eigenvalue cut, N key-frame
steps of marginalize -> [sparsify] -> solve -> write-back (tests/test_gpu_sliding_long.py::run_sequence, run = ("ora",)). The oracle needs
minutes per step at this size (cyclic Jacobi of a 915-column prior, dense-prior solves in plain C), which is why the -m gpu suite cannot
run it beside the device: it is generated once, on the CPU, and the device's sequence is compared with it step by step
(tests/test_gpu_sliding_full_size.py). Usage: python scripts/gen_sliding_golden.py [n_steps] [dense|sparsified]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle
from test_gpu_sliding_long import run_sequence

FULL = dict(n_win=12, n_kf=40, n_lmk=10200, length=20.0, keep_cap=300)

if __name__ == "__main__":
    n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    sparsif = (sys.argv[2] if len(sys.argv) > 2 else "sparsified") == "sparsified"
    rec = {"it": [], "term": [], "cost": [], "rank": [], "pose": [], "T": [], "kfs": []}
    t0 = time.time()
    def snap(step, side, st, kfs2, result, rank):
        it, term, cost, r = result
        rec["it"].append(it); rec["term"].append(term); rec["cost"].append(cost); rec["rank"].append(rank)
        rec["pose"].append(np.array(r["pose"], dtype=float).copy()); rec["T"].append(st["T"].copy()); rec["kfs"].append(np.array(kfs2))
        print(f"step {step}: it {it} term {term} cost {cost:.8f} rank {rank}  ({time.time() - t0:.0f} s)", flush=True)
    log, _, sides = run_sequence(None, oracle, True, sparsif, "reference", n_steps=n_steps, run=("ora",), snap=snap, **FULL)
    out = os.path.join(ROOT, "tests", "golden", f"sliding_config3_size_{'sparsified' if sparsif else 'dense'}.npz")
    np.savez_compressed(out, it=np.array(rec["it"]), term=np.array(rec["term"]), cost=np.array(rec["cost"]), rank=np.array(rec["rank"]),
                        pose=np.array(rec["pose"]), T=np.array(rec["T"]), kfs=np.array(rec["kfs"]), p=sides["ora"]["p"],
                        params=np.array([FULL["n_win"], FULL["n_kf"], FULL["n_lmk"], FULL["keep_cap"], n_steps]), length=FULL["length"])
    print("wrote", out, os.path.getsize(out), "bytes")
