#!/usr/bin/env python
"""Batched throughput with the windows split over two handles driven by two host threads (two HIP streams): the
single-workgroup-per-window k_solve of one half overlaps the landmark kernels of the other."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sadvio_amd import capi, synthetic

nw = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nh = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ws = [synthetic.make_window(seed=20250404 + i) for i in range(4)]
opts = capi.gn_options(10); opts.max_num_consecutive_invalid_steps = 1000
bes = []
for h in range(nh):
    be = capi.Backend(device=0, use_graph=not os.environ.get("NOGRAPH"))
    be.set_windows([ws[i % 4] for i in range(nw // nh)])
    for _ in range(2): be.solve(opts)
    bes.append(be)
reps = 6
def run(be):
    for _ in range(reps): be.solve(opts)
t = time.perf_counter()
th = [threading.Thread(target=run, args=(be,)) for be in bes]
for x in th: x.start()
for x in th: x.join()
dt = time.perf_counter() - t
print(f"{nw} windows over {nh} handle(s): {dt / reps * 1e3:.3f} ms per batch -> {nw * 10 * reps / dt:.0f} it/s")
for be in bes: be.close()
