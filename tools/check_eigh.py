"""Developer check: eigen-solver against LAPACK for a few sizes (prints errors instead of asserting)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multioptpy_amd import ops  # noqa: E402
from multioptpy_amd._lib import MoptError  # noqa: E402

rng = np.random.default_rng(1)
for n in [int(a) for a in sys.argv[1:]] or [40, 66, 100, 150, 300]:
    for B in (1, 3):
        A = rng.standard_normal((B, n, n))
        A = A + A.transpose(0, 2, 1)
        try:
            w, v = ops.eigh(A)
        except MoptError as e:
            print(n, B, "FAILED", str(e)[:80])
            continue
        err = max(np.abs(w[b] - np.linalg.eigvalsh(A[b])).max() for b in range(B))
        orth = max(np.abs(v[b].T @ v[b] - np.eye(n)).max() for b in range(B))
        print(n, B, "eig err %.2e orth %.2e" % (err, orth))

# molecular Hessian with a dense low end (large dstein clusters): the case that was slow
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import cases  # noqa: E402
from oracle import projection as pj  # noqa: E402
for N in (40, 100):
    x = cases.cluster(N, jitter=0.3) * 0.95
    A = pj.project_hessian_trrot(cases.lj(x, hessian=True)[2], x)
    A = 0.5 * (A + A.T)
    ops.eigh(A)
    t0 = time.perf_counter()
    w, v = ops.eigh(A)
    dt = time.perf_counter() - t0
    wr = np.linalg.eigvalsh(A)
    nrm = np.abs(A).sum(axis=1).max()
    print("LJ N=%d: %.2f ms, eig err %.2e (rel), orth %.2e, resid %.2e" % (N, 1e3 * dt, np.abs(w - wr).max() / nrm, np.abs(v.T @ v - np.eye(3 * N)).max(),
                                                                           np.abs(A @ v - v * w).max() / nrm))
