import os, sys, io, contextlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests/golden")
import numpy as np, cases
from multioptpy_amd.Optimizer.rsprfo import EnhancedRSPRFO
from oracle.rsprfo import EnhancedRSPRFO as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else 7
n = 3 * N
x0 = cases.cluster(N, jitter=0.3) * 0.95
H0 = cases.lj(x0, hessian=True)[2]
a = EnhancedRSPRFO(method="rsprfo_bofill", saddle_order=1, element_list=["Ar"] * N)
b = O(method="rsprfo_bofill", saddle_order=1, element_list=["Ar"] * N)
for o in (a, b):
    o.set_hessian(H0.copy()); o.set_bias_hessian(np.zeros((n, n)))
xs, px, pg, pe = x0.copy(), [], [], 0.0
for it in range(15):
    e, g = cases.lj(xs)[:2]
    ma = a.run(xs.reshape(-1, 1), g.reshape(-1, 1), pg, px, e, pe, [], [], g.reshape(-1, 1), pg)
    with contextlib.redirect_stdout(io.StringIO()):
        mb = b.run(xs.reshape(-1, 1).copy(), g.reshape(-1, 1).copy(), pg, px, e, pe, [], [], g.reshape(-1, 1).copy(), pg)
    w = np.linalg.eigvalsh(np.asarray(a.hessian))
    print("   gpu ts", a.ts_mode_idx, getattr(a, "last_info", None), "| oracle ts", b.ts_mode_idx, "|mb|", np.linalg.norm(mb))
    print(it, "dev", np.abs(np.asarray(ma) - np.asarray(mb)).max(), "|step|", np.linalg.norm(ma), "H dev", np.abs(np.asarray(a.hessian) - np.asarray(b.hessian)).max(),
          "lowest eig", w[:3], "tr", a.trust_radius, b.trust_radius)
    px, pg, pe = xs.reshape(-1, 1).copy(), g.reshape(-1, 1).copy(), e
    step = np.asarray(ma).reshape(N, 3)
    xs = xs - step * min(1.0, 0.1 / max(np.linalg.norm(step), 1e-12))
