"""Developer check: projected Hessians / gradients of the current library against another build (MOPT_HIP_LIB), bit for bit."""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
if len(sys.argv) > 1 and sys.argv[1] == "dump":
    import cases
    from multioptpy_amd import ops
    rng = np.random.default_rng(3)
    outs = []
    geoms = [cases.cluster(N) for N in (3, 7, 14, 40, 100, 113)]
    geoms.append(np.array([[0.0, 0.0, -1.1], [0.0, 0.0, 0.0], [0.0, 0.0, 1.2]]))          # linear, on an axis
    geoms.append(np.array([[0.3, 0.1, -1.1], [0.6, 0.2, 0.0], [0.9, 0.3, 1.1]]))          # linear, general direction
    geoms.append(np.array([[0.0, 0.0, 0.0], [1.0, 0.2, 0.3]]))                            # diatomic
    for x in geoms:
        n = 3 * len(x)
        H = rng.normal(size=(n, n)); H = 0.5 * (H + H.T)
        outs.append(ops.project_hess(H, x))
    np.savez(sys.argv[2], *outs)
else:
    a, b = "/tmp/cmp_new.npz", "/tmp/cmp_old.npz"
    subprocess.check_call([sys.executable, __file__, "dump", a], env=dict(os.environ, MOPT_HIP_LIB=""))
    subprocess.check_call([sys.executable, __file__, "dump", b], env=dict(os.environ, MOPT_HIP_LIB=sys.argv[1] if len(sys.argv) > 1 else "ab/oldproj.so"))
    za, zb = np.load(a), np.load(b)
    for k in za.files:
        print(k, za[k].shape, "bitwise equal" if np.array_equal(za[k], zb[k]) else f"DIFFER max {np.abs(za[k] - zb[k]).max():.3e}")
