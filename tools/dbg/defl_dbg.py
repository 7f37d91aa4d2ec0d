import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden")
import numpy as np
from multioptpy_amd import ops
rng = np.random.default_rng(9)
n = 256
B = rng.normal(size=(96, 96)); B = 0.5 * (B + B.T)
A0 = np.diag(rng.normal(size=n))
A1 = np.eye(n); A1[:96, :96] = B
A2 = np.eye(n); A2[-96:, -96:] = B
A3 = np.eye(n) * 2.5
for name, A in (("diag", A0), ("B+I", A1), ("I+B", A2), ("2.5I", A3)):
    for batch in (1, 4):
        try:
            w, v = ops.eigh(np.stack([A] * batch))
            wr = np.linalg.eigvalsh(A)
            print(name, batch, "nan" if np.isnan(w).any() else np.abs(w[0] - wr).max(), flush=True)
        except Exception as e:
            print(name, batch, "EXC", e, flush=True)
