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
