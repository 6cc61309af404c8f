"""Generates tests/golden/config3_marg_ref.npz: the ORACLE's marginalisation (oracle/marg.c, pinned on the reference's
marginalization_test.cpp fixture; Marginalization::computeSchurComplement / rankReveallingDecomposition /
computeJacobiansAndResiduals, cpp/src/optimizers/marginalization.cpp:213-265,318-342,516-530) of the config-3 sized
window of tests/test_gpu_marg.py at n = 915 (300 kept landmarks) and n = 1 215 (400): the sizes at which the device takes
its MFMA block Jacobi + register-resident pivoted Cholesky paths. ~15 s + ~35 s of CPU. Run from the repo root:

    python tests/golden/make_golden_marg.py

A full Ak at these sizes is 6.7 + 11.8 MB of incompressible doubles, so the fixture holds Ak through quantities that pin every
entry of it without storing every entry: its diagonal, its spectrum, Ak @ V for 32 seeded Gaussian probe vectors (an error E in
Ak shows as ||E V|| ~ ||E||_F sqrt(32)), every 32nd row in full, bk, and the prior's own invariants J^T J V, J^T r0, n_full.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle  # noqa: E402
from golden_util import _window_checksum, config3_marg_case, N_PROBE, ROW_STRIDE  # noqa: E402

oracle.build()
out = {}
for n_keep in (300, 400):
    w, args = config3_marg_case(n_keep)
    o = oracle.marginalize(w, **args)
    n = o["n"]
    V = np.random.default_rng(n).standard_normal((n, N_PROBE))
    Ak = o["Ak"][:n, :n]
    JtJ = o["J"].T @ o["J"]
    p = f"k{n_keep}_"
    out.update({p + "checksum": _window_checksum(w), p + "m": o["m"], p + "n": n, p + "n_full": o["n_full"], p + "kf_col": o["kf_col"],
                p + "lmk_col": o["lmk_col"], p + "Ak_diag": np.diag(Ak).copy(), p + "Ak_eig": np.linalg.eigvalsh(Ak), p + "Ak_V": Ak @ V,
                p + "Ak_rows": Ak[::ROW_STRIDE].copy(), p + "bk": o["bk"][:n].copy(), p + "JtJ_V": JtJ @ V, p + "JtJ_diag": np.diag(JtJ).copy(),
                p + "Jtr0": o["J"].T @ o["r0"]})
    print(f"n_keep {n_keep}: m {o['m']} n {n} n_full {o['n_full']} |Ak - JtJ|max/|Ak|max {np.abs(Ak - JtJ).max() / np.abs(Ak).max():.2e}")
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "config3_marg_ref.npz"), **out)
