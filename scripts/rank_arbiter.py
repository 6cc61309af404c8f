"""Which rank does the reference's criterion give when an eigenvalue of Ak sits on its absolute 1e-12 cut (marginalization.hpp:58,
marginalization.cpp:318-342)? Runs the ORACLE side of tests/test_gpu_sliding_long.py's sequences on the CPU, lists for every step the
eigenvalues of Ak nearest to the cut with the rounding-noise band n eps lambda_max of a double-precision eigen-decomposition, and for
the steps whose decision lies inside that band arbitrates in extended precision (mpmath, 50 digits): the eigenvalues of the oracle's
double-precision Ak, and of the Schur complement formed in 50 digits from the same double-precision A (computeInformationAndGradient's
output). Usage: python scripts/rank_arbiter.py [vio|vo] [dense|sparsified] > profiles/r06_rank_arbiter.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle   # noqa: E402
import test_gpu_sliding_long as T   # noqa: E402

CUT = 1e-12
EPS = 2.220446049250313e-16


def main():
    vio = (sys.argv[1] if len(sys.argv) > 1 else "vio") == "vio"
    sparsif = (sys.argv[2] if len(sys.argv) > 2 else "dense") == "sparsified"
    oracle.build()
    rows = []

    def hook(step, side, w, g, args):
        Ak = g["Ak"][: g["n"], : g["n"]]
        ev = np.linalg.eigvalsh(0.5 * (Ak + Ak.T))
        band = g["n"] * EPS * ev.max()
        near = ev[np.argsort(np.abs(ev - CUT))[:3]]
        rows.append(dict(step=step, n=int(g["n"]), n_full=int(g["n_full"]), lmax=float(ev.max()), band=float(band), near=[float(x) for x in near],
                         lapack_rank=int((ev > CUT).sum()), Ak=Ak.copy() if abs(near[0] - CUT) <= 50 * band else None, args=args, w=w))

    T.run_sequence(None, oracle, vio, sparsif, "reference", run=("ora",), hook=hook)
    print(f"# oracle side of the {'VIO' if vio else 'VO'} {'sparsified' if sparsif else 'dense'} sliding sequence, reference cut 1e-12")
    print("# step  n  n_full(oracle, cyclic Jacobi)  rank(LAPACK eigvalsh of the same Ak)  lambda_max  noise band n eps lambda_max  eigenvalues nearest the cut")
    for r in rows:
        flag = "  <-- decision inside the noise band" if abs(r["near"][0] - CUT) <= r["band"] else ""
        print(f"{r['step']:3d} {r['n']:4d} {r['n_full']:4d} {r['lapack_rank']:4d}  {r['lmax']:.3e}  {r['band']:.2e}  " + " ".join(f"{x:+.3e}" for x in r["near"]) + flag)
    import mpmath as mp
    mp.mp.dps = 50
    for r in rows:
        if r["Ak"] is None or abs(r["near"][0] - CUT) > r["band"]:
            continue
        n = r["n"]
        A = mp.matrix(n, n)
        for i in range(n):
            for j in range(n):
                A[i, j] = mp.mpf(0.5) * (mp.mpf(float(r["Ak"][i, j])) + mp.mpf(float(r["Ak"][j, i])))
        ev = sorted(mp.eigsy(A, eigvals_only=True), key=lambda x: abs(x - mp.mpf(CUT)))
        k50 = sum(1 for x in mp.eigsy(A, eigvals_only=True) if x > mp.mpf(CUT))
        print(f"# step {r['step']}: 50-digit eigenvalues of the oracle's double-precision Ak nearest the cut: " + " ".join(mp.nstr(x, 6) for x in ev[:3]) + f"  -> rank {k50} of {n}")
        # the same Schur complement formed in 50 digits from the double-precision A (un-reduced information of the step)
        g = oracle.marginalize(r["w"], want_full=True, **r["args"])
        Af, m = g["A_full"], g["m"]
        N = m + n
        Amm = mp.matrix(m, m); Arm = mp.matrix(n, m); Arr = mp.matrix(n, n)
        for i in range(m):
            for j in range(m):
                Amm[i, j] = mp.mpf(0.5) * (mp.mpf(float(Af[i, j])) + mp.mpf(float(Af[j, i])))
        for i in range(n):
            for j in range(m):
                Arm[i, j] = mp.mpf(float(Af[m + i, j]))
            for j in range(n):
                Arr[i, j] = mp.mpf(float(Af[m + i, m + j]))
        E, Q = mp.eigsy(Amm)
        inv = mp.matrix(m, m)
        for k in range(m):
            if E[k] > mp.mpf(CUT):
                for i in range(m):
                    for j in range(m):
                        inv[i, j] += Q[i, k] * Q[j, k] / E[k]
        Ake = Arr - Arm * inv * Arm.T
        Ake = (Ake + Ake.T) * mp.mpf(0.5)
        eve = mp.eigsy(Ake, eigvals_only=True)
        ke = sum(1 for x in eve if x > mp.mpf(CUT))
        near = sorted(eve, key=lambda x: abs(x - mp.mpf(CUT)))[:3]
        print(f"#          Schur complement in 50 digits from the double-precision A: nearest " + " ".join(mp.nstr(x, 6) for x in near) + f"  -> rank {ke} of {n};"
              f" |Ak_double - Ak_50| max {mp.nstr(max(abs(Ake[i, j] - A[i, j]) for i in range(n) for j in range(n)), 3)}")


if __name__ == "__main__":
    main()
