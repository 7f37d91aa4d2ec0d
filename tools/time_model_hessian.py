"""Developer timing: model Hessians through the plugin class (host arrays in and out) -- one geometry and a 32-image batch, the C30H62
chain of tests/test_gpu_configs.py (92 atoms: thousands of internal coordinates) and argon clusters (pair terms only).
`--reference` times the reference's own classes instead (build container only: /root/reference must exist; CPU, Python loops)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases  # noqa: E402


def chain():
    rng = np.random.default_rng(3)
    el, xyz = [], []
    for k in range(30):
        c = np.array([1.26 * k, 0.45 * (-1) ** k, 0.0])
        el.append("C"); xyz.append(c)
        for s in (-1.0, 1.0):
            el.append("H"); xyz.append(c + np.array([0.0, 0.62 * (-1) ** k, 0.89 * s]))
    el += ["H", "H"]
    xyz += [xyz[0] + np.array([-0.9, 0.62, 0.0]), xyz[87] + np.array([0.9, 0.62 * (-1) ** 29, 0.0])]
    return el, (np.array(xyz) + rng.normal(0, 0.02, (len(xyz), 3))) / cases.BOHR2ANG


systems = [("C30H62", *chain()), ("Ar100", ["Ar"] * 100, cases.cluster(100)), ("Ar200", ["Ar"] * 200, cases.cluster(200))]
models = ("fischerd3old", "fischerd3", "fischer", "schlegel", "morse")

if "--reference" in sys.argv:
    import types
    pkg = types.ModuleType("multioptpy"); pkg.__path__ = ["/root/reference/multioptpy"]; sys.modules["multioptpy"] = pkg
    sys.dont_write_bytecode = True
    import contextlib, io
    from multioptpy.ModelHessian.approx_hessian import ApproxHessian
    for tag, el, x in systems[:2]:
        for m in models:
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                ApproxHessian().main(x.copy(), list(el), np.zeros_like(x), approx_hess_type=m)
            print(f"reference {tag:7s} {m:13s} {time.perf_counter() - t0:8.3f} s per geometry (this container's CPU)", flush=True)
    sys.exit(0)

from multioptpy_amd.ModelHessian.approx_hessian import ApproxHessian  # noqa: E402
for tag, el, x in systems:
    for m in models:
        AH = ApproxHessian()
        AH.main(x, el, None, m)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); AH.main(x, el, None, m); ts.append(time.perf_counter() - t0)
        X32 = np.array([x + 0.001 * k for k in range(32)])
        AH.batch(X32, el, m)
        tb = []
        for _ in range(3):
            t0 = time.perf_counter(); AH.batch(X32, el, m); tb.append(time.perf_counter() - t0)
        print(f"device {tag:7s} {m:13s} one geometry {1e3 * np.median(ts):8.3f} ms   32 images {1e3 * np.median(tb):8.3f} ms (host arrays in and out)", flush=True)
