"""Phase timings of k_small_rsprfo (MOPT_SMALL_DBG=1) on the LJ cluster trajectory of bench.py --single-geometry."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import numpy as np, cases
from multioptpy_amd.Optimizer.rsprfo import EnhancedRSPRFO
N = int(sys.argv[1]) if len(sys.argv) > 1 else 7
n = 3 * N
x = cases.cluster(N, jitter=0.3) * 0.95
opt = EnhancedRSPRFO(method="rsprfo_bofill", saddle_order=1, element_list=["Ar"] * N)
opt.set_hessian(cases.lj(x, hessian=True)[2].copy()); opt.set_bias_hessian(np.zeros((n, n)))
xs = x.copy(); pg = px = []; pe = 0.0
for it in range(8):
    e, g = cases.lj(xs)[:2]
    mv = opt.run(xs.reshape(-1, 1), g.reshape(-1, 1), pg, px, e, pe, [], [], g.reshape(-1, 1), pg)
    px, pg, pe = xs.reshape(-1, 1).copy(), g.reshape(-1, 1).copy(), e
    step = np.asarray(mv).reshape(N, 3)
    xs = xs - step * min(1.0, 0.1 / max(np.linalg.norm(step), 1e-12))
