"""Where the time of one RSIRFO.run goes: the whole call, the C-ABI call inside it, the kernel phases (MOPT_SMALL_DBG=1 prints them)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import numpy as np, cases
from multioptpy_amd import _lib
from multioptpy_amd.Optimizer import rsirfo as mod
N = int(sys.argv[1]) if len(sys.argv) > 1 else 7
n = 3 * N
x = cases.cluster(N, jitter=0.3) * 0.95
opt = mod.RSIRFO(method="rsirfo_bfgs", saddle_order=0, element_list=["Ar"] * N)
opt.set_hessian(cases.lj(x, hessian=True)[2].copy()); opt.set_bias_hessian(np.zeros((n, n)))
real = _lib.lib().mopt_rsirfo_run
inner = []
class L:
    def __getattr__(self, k):
        f = getattr(_lib.lib(), k)
        if k != "mopt_rsirfo_run":
            return f
        def g(*a):
            t0 = time.perf_counter(); r = f(*a); inner.append(time.perf_counter() - t0); return r
        return g
mod.lib = lambda: L()
xs = x.copy(); pg = px = []
outer = []
for it in range(60):
    e, g = cases.lj(xs)[:2]
    t0 = time.perf_counter()
    mv = opt.run(xs.reshape(-1, 1), g.reshape(-1, 1), pg, px, e, 0.0, [], [], g.reshape(-1, 1), pg)
    outer.append(time.perf_counter() - t0)
    px, pg = xs.reshape(-1, 1).copy(), g.reshape(-1, 1).copy()
    step = np.asarray(mv).reshape(N, 3)
    xs = xs - step * min(1.0, 0.1 / max(np.linalg.norm(step), 1e-12))
print("n = %d: run() median %.1f us, C-ABI call inside %.1f us" % (n, 1e6 * np.median(outer[10:]), 1e6 * np.median(inner[10:])))
