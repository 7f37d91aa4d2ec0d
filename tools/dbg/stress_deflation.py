"""One-off stress of the early-stopping tridiagonalisation (developer tool): identity / scaled-identity / diagonal + low rank of random rank,
scale and noise, mixed in one batch, against LAPACK (eigenvalues 1e-13 ||A||, orthogonality 1e-11, residual 5e-12 ||A||)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from multioptpy_amd import ops  # noqa: E402
import test_gpu_configs as t  # noqa: E402

rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
nfail = 0
for trial in range(int(os.environ.get("TRIALS", "40"))):
    n = int(rng.choice([40, 64, 100, 120, 200, 256, 300, 333, 384, 450, 600]))
    batch = int(rng.choice([1, 3, 8, 16, 32]))
    if n * n * batch > 3.0e6:
        batch = max(1, int(3.0e6 // (n * n)))
    A = []
    for b in range(batch):
        scale = 10.0 ** rng.uniform(-6, 6)
        r = int(rng.integers(0, min(n - 2, 90)))
        kind = rng.integers(0, 4)
        if kind == 0:
            D = np.ones(n)
        elif kind == 1:
            D = np.full(n, 10.0 ** rng.uniform(-4, 2))
        elif kind == 2:
            D = np.where(np.arange(n) < n // 2, 1.0, 3.0)      # two-level diagonal: does not deflate after r steps
        else:
            D = rng.uniform(0.5, 2.0, n)
        U = rng.normal(size=(n, r)) * 10.0 ** rng.uniform(-2, 0.5)
        S = rng.normal(size=(r, r)); S = 0.5 * (S + S.T)
        M = np.diag(D) + U @ S @ U.T
        lvl = 10.0 ** rng.uniform(-19, -11) if rng.random() < 0.5 else 0.0
        N = rng.normal(size=(n, n)) * lvl
        M = (M + 0.5 * (N + N.T)) * scale
        A.append(0.5 * (M + M.T))
    A = np.stack(A)
    try:
        t._check_eigh(ops, A)
    except AssertionError as e:
        nfail += 1
        print("FAIL trial", trial, "n", n, "batch", batch, str(e)[:200])
print("stress done:", nfail, "failures")
