"""A 64-window plain batch (the bench's batched leg) solved a few times: the command scripts/prof_batched.sh profiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sadvio_amd import capi, synthetic
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ws = [synthetic.make_window(seed=20250404 + 100 + i) for i in range(8)]
ws = [ws[i % 8] for i in range(nw)]
opts = capi.gn_options(10); opts.max_num_consecutive_invalid_steps = 1000
be = capi.Backend(device=0)
be.set_windows(ws)
for _ in range(3):
    s = be.solve(opts)
print("final cost", s[0].final_cost, "iterations", sum(x.iterations for x in s))
be.close()
