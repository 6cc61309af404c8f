import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from sadvio_amd import capi
from vio_helpers import make_vio_window
opts = capi.gn_options(10); opts.max_num_consecutive_invalid_steps = 1000
ws = [make_vio_window(n_kf=12, n_lmk=7200, seed=6 + i) for i in range(4)]
for nw in (1, 2, 4, 8, 16):
    be = capi.Backend(device=0, use_graph=True)
    be.set_windows([ws[i % 4] for i in range(nw)])
    for _ in range(3): be.solve(opts)
    t = time.perf_counter()
    for _ in range(10): s = be.solve(opts)
    dt = (time.perf_counter() - t) / 10
    print(f"PF_WG={os.environ.get('SADVIO_PF_WG','auto')} vio windows {nw}: {dt*1e3:.3f} ms -> {nw*10/dt:.0f} it/s", flush=True)
    be.close()
