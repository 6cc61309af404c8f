import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sadvio_amd import capi, synthetic
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 1
factor = capi.FACTOR_ANGULAR if os.environ.get("SADVIO_ANGULAR") else capi.FACTOR_PIXEL   # the reference's shipped config uses the angular backend
ws = [synthetic.make_window(seed=20250404 + i, factor=factor) for i in range(min(nw, 4))]
ws = [ws[i % len(ws)] for i in range(nw)]
opts = capi.gn_options(10); opts.max_num_consecutive_invalid_steps = 1000
be = capi.Backend(device=0, profile_kernels=True)
be.set_windows(ws)
for _ in range(3): be.solve(opts)
be.set_windows(ws)
t=time.perf_counter()
for _ in range(10): s = be.solve(opts)
dt=(time.perf_counter()-t)/10
kt = be.kernel_times()
print(f"debug={os.environ.get('SADVIO_DEBUG','0')} windows={nw} wall/solve {dt*1e3:.3f} ms (profiled) ", {k: round(v['avg_us'],2) for k,v in kt.items()}, "final cost", s[0].final_cost)
be.close()
be = capi.Backend(device=0)
be.set_windows(ws)
for _ in range(3): be.solve(opts)
t=time.perf_counter()
for _ in range(20): s = be.solve(opts)
dt=(time.perf_counter()-t)/20
print(f"   unprofiled wall/solve {dt*1e3:.3f} ms -> {nw*10/dt:.0f} it/s")
t=time.perf_counter()
for _ in range(5):
    be.set_windows(ws)
print(f"   set_windows wall {1e3*(time.perf_counter()-t)/5:.3f} ms (flatten done; pageable host -> device upload + tiling), {sum(w.n_obs for w in ws)} obs")
be.close()
be = capi.Backend(device=0, use_graph=True)
be.set_windows(ws)
for _ in range(3): be.solve(opts)
t=time.perf_counter()
for _ in range(20): s = be.solve(opts)
dt=(time.perf_counter()-t)/20
print(f"   hipGraph   wall/solve {dt*1e3:.3f} ms -> {nw*10/dt:.0f} it/s   final cost {s[0].final_cost}")
be = capi.Backend(device=0, use_graph=True)
be.set_windows(ws)
for _ in range(3): be.solve(opts); be.get_deltas(0)
t=time.perf_counter()
for _ in range(30):
    be.set_windows(ws); be.solve(opts); d = be.get_deltas(0)
dt=(time.perf_counter()-t)/30
t=time.perf_counter()
for _ in range(30): d = be.get_deltas(0)
dg=(time.perf_counter()-t)/30
be.solve(opts)
t=time.perf_counter(); d = be.get_deltas(0); dg1=time.perf_counter()-t
print(f"   upload-inclusive (set_windows + solve + get_deltas) {dt*1e3:.3f} ms -> {nw*10/dt:.0f} it/s; get_deltas first {dg1*1e6:.0f} us, cached {dg*1e6:.0f} us")
