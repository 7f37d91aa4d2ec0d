import numpy as np, os, sys
sys.path.insert(0, os.getcwd())
from multioptpy_amd import ops
rng = np.random.default_rng(0)
n = int(os.environ.get("N", "300")); B = int(os.environ.get("B", "32"))
A = rng.standard_normal((B, n, n)); A = A + A.transpose(0, 2, 1)
for it in range(3):
    w, v = ops.eigh(A)
print("ok", np.abs(w[0] - np.linalg.eigvalsh(A[0])).max())
