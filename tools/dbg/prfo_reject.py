import os, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden"))
import numpy as np, cases
from multioptpy_amd.Optimizer.rsprfo import EnhancedRSPRFO
from oracle.rsprfo import EnhancedRSPRFO as Oracle
natoms=7; n=21
x0 = cases.cluster(natoms, jitter=0.05) * 0.99
v = np.random.default_rng(natoms).normal(size=n); v /= np.linalg.norm(v)
H0 = cases.lj(x0, hessian=True)[2] + 3.0e6 * np.outer(v, v)
a = EnhancedRSPRFO(method="rsprfo_bofill", saddle_order=1, element_list=["Ar"] * natoms)
b = Oracle(method="rsprfo_bofill", saddle_order=1, element_list=["Ar"] * natoms)
for o in (a, b):
    o.set_hessian(H0.copy()); o.set_bias_hessian(np.zeros((n, n)))
xs, px, pg, pe = x0.copy(), [], [], 0.0
for it in range(3):
    e, g = cases.lj(xs)[:2]
    ma = a.run(xs.reshape(-1, 1), g.reshape(-1, 1), pg, px, e, pe, [], [], g.reshape(-1, 1), pg)
    with contextlib.redirect_stdout(io.StringIO()):
        mb = b.run(xs.reshape(-1, 1).copy(), g.reshape(-1, 1).copy(), pg, px, e, pe, [], [], g.reshape(-1, 1).copy(), pg)
    print(it, b.trace.get('basis_dependent'), np.abs(np.asarray(ma)-np.asarray(mb)).max(), np.linalg.norm(ma), np.linalg.norm(mb), a.trust_radius, b.trust_radius, a.last_info, b.trace.get("shifted"), np.abs(np.asarray(a.hessian)-np.asarray(b.hessian)).max())
    px, pg, pe = xs.reshape(-1, 1).copy(), g.reshape(-1, 1).copy(), e
    step = np.asarray(ma).reshape(natoms, 3)
    xs = xs - step * min(1.0, 0.1 / max(np.linalg.norm(step), 1e-12))
