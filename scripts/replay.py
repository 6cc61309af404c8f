"""Replay dumped windows (SADVIOW1 files, include/sadvio_io.hpp / sadvio_amd/io.py) through the GPU library and, when
it is built, the CPU oracle — same inputs, same options — and report the solve summaries and the GPU-vs-CPU parity.

    python scripts/replay.py [--iters 20] [--no-oracle] window0.sadvio window1.sadvio ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sadvio_amd import capi, io  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="+")
    ap.add_argument("--iters", type=int, default=20, help="max_num_iterations (the reference: 20, AOptimizer.cpp:318)")
    ap.add_argument("--huber", type=float, default=0.0, help="ceres::HuberLoss parameter (0 = none)")
    ap.add_argument("--no-oracle", action="store_true")
    a = ap.parse_args()
    oracle = None
    if not a.no_oracle:
        try:
            from oracle import oracle as _o
            _o.lib()
            oracle = _o
        except Exception as e:  # the oracle is test infrastructure: replay still runs the GPU path without it
            print(f"# oracle unavailable ({e}); GPU only", file=sys.stderr)
    opts = capi.reference_options(); opts.max_num_iterations = a.iters; opts.huber_a = a.huber
    be = capi.Backend(device=0)
    for path in a.files:
        w = io.load_window(path)
        t0 = time.perf_counter()
        be.set_windows([w])
        s = be.solve(opts)[0]
        d = be.get_deltas(0)
        ms = 1e3 * (time.perf_counter() - t0)
        rec = {"file": os.path.basename(path), "n_kf": w.n_kf, "n_lmk": w.n_lmk, "n_obs": w.n_obs, "has_imu": w.has_imu,
               "gpu": {"initial_cost": s.initial_cost, "final_cost": s.final_cost, "iterations": s.iterations, "termination": s.termination,
                       "ms_upload_solve_readback": round(ms, 3)}}
        if oracle:
            t0 = time.perf_counter()
            ref = oracle.solve(w, opts)
            rs = ref["summary"]
            rec["cpu_oracle"] = {"initial_cost": rs.initial_cost, "final_cost": rs.final_cost, "iterations": rs.iterations,
                                 "termination": rs.termination, "ms": round(1e3 * (time.perf_counter() - t0), 3)}
            rec["parity"] = {"max_pose_delta_diff": float(np.abs(d["pose"] - ref["pose"]).max()) if w.n_kf else 0.0,
                             "max_lmk_delta_diff": float(np.abs(d["lmk"] - ref["lmk"]).max()) if w.n_lmk else 0.0,
                             "rel_final_cost_diff": abs(s.final_cost - rs.final_cost) / max(abs(rs.final_cost), 1e-300)}
        print(json.dumps(rec))
    be.close()


if __name__ == "__main__":
    main()
