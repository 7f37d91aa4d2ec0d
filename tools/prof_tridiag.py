"""Phase profile of the persistent tridiagonalisation kernel (developer tool, not part of the product path).

    MOPT_TRI_PROF=1 MOPT_EIGH_ALGO=tridiag python tools/prof_tridiag.py

prints the per-step time k_tri_persist spends polling partner workgroups, in the reflector prologue, in the LDS main
pass and at its barrier (wall_clock64 stamps taken by thread 0 of every workgroup, averaged)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multioptpy_amd import ops  # noqa: E402

rng = np.random.default_rng(0)
n = int(os.environ.get("N", "300"))
B = int(os.environ.get("B", "32"))
A = rng.standard_normal((B, n, n))
A = A + A.transpose(0, 2, 1)
for it in range(3):
    w, v = ops.eigh(A)
print("max |w - LAPACK| of matrix 0:", np.abs(w[0] - np.linalg.eigvalsh(A[0])).max())
