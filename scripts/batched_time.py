"""Wall time of a plain batch of distinct windows (the bench's batched leg, without the rest of bench.py): python scripts/batched_time.py [windows] [repeats]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sadvio_amd import capi, synthetic
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 30
factor = capi.FACTOR_ANGULAR if os.environ.get("SADVIO_ANGULAR") else capi.FACTOR_PIXEL
base = [synthetic.make_window(seed=20250404 + 100 + i, factor=factor) for i in range(min(nw, 16))]
ws = [base[i % len(base)] for i in range(nw)]
opts = capi.gn_options(10); opts.max_num_consecutive_invalid_steps = 1000
be = capi.Backend(device=0)
be.set_windows(ws)
for _ in range(3):
    s = be.solve(opts)
best = 1e9; tot = 0.0
for _ in range(rep):
    t = time.perf_counter(); s = be.solve(opts); dt = time.perf_counter() - t
    best = min(best, dt); tot += dt
it = sum(x.iterations for x in s)
print(f"windows {nw}: mean {tot / rep * 1e3:.3f} ms -> {it / (tot / rep):.0f} it/s, best {best * 1e3:.3f} ms -> {it / best:.0f} it/s, final cost {s[0].final_cost!r}")
be.close()
