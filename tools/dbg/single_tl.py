import os, sys, time, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests/golden")
import numpy as np, cases
from multioptpy_amd import _lib
from multioptpy_amd.Optimizer.rsirfo import RSIRFO
N = int(sys.argv[1]) if len(sys.argv) > 1 else 11
n = 3 * N
x = cases.cluster(N, jitter=0.3) * 0.95
opt = RSIRFO(method="rsirfo_bfgs", saddle_order=0, element_list=["Ar"] * N)
H = cases.lj(x, hessian=True)[2]
opt.set_hessian(H.copy()); opt.set_bias_hessian(np.zeros((n, n)))
xs = x.copy(); pg = px = []
c = _lib.ctx()
for it in range(6):
    e, g = cases.lj(xs)[:2]
    if it == 4:
        _lib.check(_lib.lib().mopt_ctx_timing(c, 1), c)
    t0 = time.perf_counter()
    mv = opt.run(xs.reshape(-1, 1), g.reshape(-1, 1), pg, px, e, 0.0, [], [], g.reshape(-1, 1), pg)
    dt = time.perf_counter() - t0
    px, pg = xs.reshape(-1, 1).copy(), g.reshape(-1, 1).copy()
    step = np.asarray(mv).reshape(N, 3)
    xs = xs - step * min(1.0, 0.1 / max(np.linalg.norm(step), 1e-12))
    print("call", it, "%.3f ms" % (dt * 1e3))
ms = C.c_double(0); nl = C.c_int64(0)
os.environ["MOPT_TIMELINE_ALL"] = "1"
_lib.lib().mopt_ctx_timing_read(c, b"timeline", C.byref(ms), C.byref(nl))
