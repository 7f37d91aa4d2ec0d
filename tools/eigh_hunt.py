"""developer aid: batches of identity + low-rank matrices through mopt_eigh, checked against LAPACK"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multioptpy_amd import ops
from multioptpy_amd._lib import MoptError
n = int(os.environ.get("N", "120")); B = int(os.environ.get("B", "8"))
rng = np.random.default_rng(3)
bad = 0
for it in range(int(os.environ.get("REPS", "40"))):
    A = np.empty((B, n, n))
    for b in range(B):
        r = 8 + (it + b) % 10
        U = rng.normal(size=(n, r))
        A[b] = np.eye(n) + 0.3 * (U * rng.normal(size=r)) @ U.T / n
    try:
        w, v = ops.eigh(A)
    except MoptError as e:
        print("iter", it, "ERROR", str(e)[:100]); bad += 1; continue
    for b in range(B):
        wr = np.linalg.eigvalsh(A[b])
        e1 = np.abs(w[b] - wr).max(); e2 = np.abs(v[b].T @ v[b] - np.eye(n)).max(); e3 = np.abs(A[b] @ v[b] - v[b] * w[b]).max()
        if not (e1 < 1e-12 and e2 < 1e-11 and e3 < 1e-11):
            print("iter", it, "matrix", b, "eig %.2e orth %.2e resid %.2e" % (e1, e2, e3)); bad += 1
print("bad:", bad)
