import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sadvio_amd import capi, synthetic
from oracle import oracle
w = synthetic.make_window(n_kf=6, n_lmk=400, seed=7, factor=capi.FACTOR_ANGULAR)
opts = capi.gn_options(10)
be = capi.Backend(device=0); be.set_windows([w]); s = be.solve(opts)[0]; d = be.get_deltas(0)
ref = oracle.solve(w, opts)
print(s.as_dict()); print(ref['summary'].as_dict())
np.set_printoptions(linewidth=200)
print(ref['log'])
print(np.abs(d['pose']-ref['pose']).max(), np.abs(d['lmk']-ref['lmk']).max())
