"""Developer timing: one RSIRFO.run / EnhancedRSPRFO.run call through the plugin class (host ndarrays in and out, i.e. PCIe
and allocation included) for the single-geometry sizes of BASELINE.json's optmain configurations."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases  # noqa: E402
from multioptpy_amd.Optimizer.rsirfo import RSIRFO  # noqa: E402
from multioptpy_amd.Optimizer.rsprfo import EnhancedRSPRFO  # noqa: E402

SIZES = tuple(int(a) for a in sys.argv[1:]) or (7, 11, 22, 100, 200)
for N in SIZES:
    n = 3 * N
    x = cases.cluster(N, jitter=0.3) * 0.95
    for cls, method, order in ((RSIRFO, "rsirfo_bfgs", 0), (EnhancedRSPRFO, "rsprfo_bofill", 1)):
        opt = cls(method=method, saddle_order=order, element_list=["Ar"] * N)
        H = cases.lj(x, hessian=True)[2]
        opt.set_hessian(H.copy())
        opt.set_bias_hessian(np.zeros((n, n)))
        xs = x.copy()
        pg = px = []
        ts = []
        for it in range(8):
            e, g = cases.lj(xs)[:2]
            t0 = time.perf_counter()
            mv = opt.run(xs.reshape(-1, 1), g.reshape(-1, 1), pg, px, e, 0.0, [], [], g.reshape(-1, 1), pg)
            ts.append(time.perf_counter() - t0)
            px, pg = xs.reshape(-1, 1).copy(), g.reshape(-1, 1).copy()
            step = np.asarray(mv).reshape(N, 3)
            nrm = np.linalg.norm(step)
            xs = xs - step * min(1.0, 0.1 / max(nrm, 1e-12))
        print(f"N={N:4d} n={n:4d} {cls.__name__:15s} median {1e3 * np.median(ts[2:]):7.3f} ms per run (min {1e3 * min(ts[2:]):.3f})")
