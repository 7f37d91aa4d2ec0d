"""Kernel sequence of one RSIRFO.run / EnhancedRSPRFO.run call (run under rocprofv3 --kernel-trace; tools/dbg/trace_last_call.py reads the csv)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import numpy as np, cases
from multioptpy_amd.Optimizer.rsirfo import RSIRFO
from multioptpy_amd.Optimizer.rsprfo import EnhancedRSPRFO
N = int(sys.argv[1]) if len(sys.argv) > 1 else 7
which = sys.argv[2] if len(sys.argv) > 2 else "rsirfo"
n = 3 * N
x = cases.cluster(N, jitter=0.3) * 0.95
if which == "rsirfo":
    opt = RSIRFO(method="rsirfo_bfgs", saddle_order=0, element_list=["Ar"] * N)
else:
    opt = EnhancedRSPRFO(method="rsprfo_bofill", saddle_order=1, element_list=["Ar"] * N)
opt.set_hessian(cases.lj(x, hessian=True)[2].copy()); opt.set_bias_hessian(np.zeros((n, n)))
xs = x.copy(); pg = px = []
for it in range(12):
    e, g = cases.lj(xs)[:2]
    t0 = time.perf_counter()
    mv = opt.run(xs.reshape(-1, 1), g.reshape(-1, 1), pg, px, e, 0.0, [], [], g.reshape(-1, 1), pg)
    dt = time.perf_counter() - t0
    px, pg = xs.reshape(-1, 1).copy(), g.reshape(-1, 1).copy()
    step = np.asarray(mv).reshape(N, 3)
    xs = xs - step * min(1.0, 0.1 / max(np.linalg.norm(step), 1e-12))
    print("call", it, "%.3f ms" % (dt * 1e3), flush=True)
    time.sleep(0.01)
