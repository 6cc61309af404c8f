"""Ad-hoc GPU check: parity vs oracle + per-kernel timings on the C2 window."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sadvio_amd import capi, synthetic
from oracle import oracle

def parity(w, opts, tag):
    be = capi.Backend(device=0, profile_kernels=True)
    be.set_windows([w])
    t = time.time(); s = be.solve(opts)[0]; t1 = time.time() - t
    t = time.time(); s = be.solve(opts)[0]; t2 = time.time() - t
    d = be.get_deltas(0)
    ref = oracle.solve(w, opts)
    rs = ref["summary"]
    print(f"[{tag}] gpu: it={s.iterations} ok={s.num_successful_steps} term={s.termination} cost {s.initial_cost:.6f}->{s.final_cost:.6f} r={s.final_radius:.4g}  wall {t1*1e3:.2f} / {t2*1e3:.2f} ms")
    print(f"[{tag}] cpu: it={rs.iterations} ok={rs.num_successful_steps} term={rs.termination} cost {rs.initial_cost:.6f}->{rs.final_cost:.6f} r={rs.final_radius:.4g}")
    print(f"[{tag}] max|dpose| {np.abs(d['pose']-ref['pose']).max():.3e}  max|dlmk| {np.abs(d['lmk']-ref['lmk']).max():.3e}")
    print(f"[{tag}] kernels", be.kernel_times())
    be.close()

w = synthetic.make_window(n_kf=6, n_lmk=400, seed=7)
be = capi.Backend(device=0)
be.set_windows([w])
r, Jp, Jl = be.linearize(0)
ro, Jpo, Jlo, v = oracle.linearize(w)
print("lin r", np.abs(r-ro).max(), "Jp", np.abs(Jp-Jpo).max()/np.abs(Jpo).max(), "Jl", np.abs(Jl-Jlo).max()/np.abs(Jlo).max())
rng = np.random.default_rng(0)
pd = 0.01*rng.standard_normal((w.n_kf,6)); ld = 0.02*rng.standard_normal((w.n_lmk,3))
r, Jp, Jl = be.linearize(0, pd, ld)
ro, Jpo, Jlo, v = oracle.linearize(w, pd, ld)
print("lin@delta r", np.abs(r-ro).max(), "Jp", np.abs(Jp-Jpo).max()/np.abs(Jpo).max(), "Jl", np.abs(Jl-Jlo).max()/np.abs(Jlo).max())
be.close()
parity(w, capi.reference_options(), "small ref")
parity(w, capi.gn_options(10), "small gn10")
wa = synthetic.make_window(n_kf=6, n_lmk=400, seed=7, factor=capi.FACTOR_ANGULAR)
parity(wa, capi.reference_options(), "small angular")
w2 = synthetic.make_window()
parity(w2, capi.reference_options(), "C2 ref")
parity(w2, capi.gn_options(10), "C2 gn10")
