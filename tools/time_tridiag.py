"""Per-family kernel times of the batched eigen-solve on random symmetric matrices (developer tool).

    N=300 B=32 [MOPT_HIP_LIB=ab/<variant>.so] python tools/time_tridiag.py

Uses the context's HIP-event timing (mopt_ctx_timing) and prints the mean ms per call of every kernel family.  Works with
the timing-ablation builds of k_tri_persist3 as well: their results are garbage, an error return is reported and ignored."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multioptpy_amd import _lib  # noqa: E402

n = int(os.environ.get("N", "300"))
B = int(os.environ.get("B", "32"))
reps = int(os.environ.get("REPS", "10"))
rng = np.random.default_rng(0)
kind = os.environ.get("KIND", "random")              # random | identity | lowrank:<r> (identity + rank r: the early stop's regime)
if kind == "identity":
    A = np.stack([np.eye(n)] * B)
elif kind.startswith("lowrank:"):
    r = int(kind.split(":")[1])
    U = rng.standard_normal((B, n, r)) * 0.1
    A = np.eye(n)[None] + U @ U.transpose(0, 2, 1)
else:
    A = rng.standard_normal((B, n, n))
    A = A + A.transpose(0, 2, 1)
A = np.ascontiguousarray(A)
W = np.empty((B, n))
V = np.empty_like(A)
sw = np.zeros(B, dtype=np.int32)
lib, c = _lib.lib(), _lib.ctx()
DP, IP = C.POINTER(C.c_double), C.POINTER(C.c_int)
args = (c, n, B, A.ctypes.data_as(DP), W.ctypes.data_as(DP), V.ctypes.data_as(DP), sw.ctypes.data_as(IP))
rcs = set()
for _ in range(3):
    rcs.add(lib.mopt_eigh(*args))
lib.mopt_ctx_timing(c, 1)
for _ in range(reps):
    rcs.add(lib.mopt_eigh(*args))
out = []
for name in ("eigh_tridiag", "eigh_bisect", "eigh_invit", "eigh_backtrans", "eigh_final"):
    ms, nl = C.c_double(0.0), C.c_int64(0)
    lib.mopt_ctx_timing_read(c, name.encode(), C.byref(ms), C.byref(nl))
    out.append(f"{name} {ms.value / reps:.4f}")
lib.mopt_ctx_timing(c, 0)
print(f"n={n} B={B} {kind} rc={sorted(rcs)}  " + "  ".join(out))
