"""In-kernel phase timestamps of the throughput kernels (TS build, SADVIO_DEBUG=4096): python scripts/lm_ts.py [windows]"""
import os, sys
os.environ["SADVIO_DEBUG"] = "4096"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sadvio_amd import capi, synthetic
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ws = [synthetic.make_window(seed=20250404 + 100 + i) for i in range(8)]
ws = [ws[i % 8] for i in range(nw)]
opts = capi.gn_options(10); opts.max_num_consecutive_invalid_steps = 1000
be = capi.Backend(device=0)
be.set_windows(ws)
for _ in range(2):
    be.solve(opts)
be.close()
