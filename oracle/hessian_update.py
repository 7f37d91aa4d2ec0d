"""Oracle: quasi-Newton Hessian update rules (test infrastructure, see oracle/__init__.py).

Restates, in dense NumPy, what the reference computes in
  multioptpy/Optimizer/hessian_update.py:35-433        (ModelHessianUpdate)
  multioptpy/Optimizer/block_hessian_update.py:75-709  (BlockHessianUpdate, effective q = 1)
and the substring dispatch of multioptpy/Optimizer/rsirfo.py:208-251,1341-1353.

All vectors are 1-D here (the reference carries (n,1) columns).  Every function
returns delta_H such that the caller does ``H += delta_H`` then symmetrises.
"""
import numpy as np

DENOM = 1e-10        # hessian_update.py:25
BLK_DENOM = 1e-12    # block_hessian_update.py:24
INV_REG = 1e-10      # block_hessian_update.py:24


# ----------------------------------------------------------------------------
# standard rules (hessian_update.py)
# ----------------------------------------------------------------------------
def bfgs_delta(H, s, y):
    """hessian_update.py:35-65.  (Hs)(s^T H^T) == outer(Hs, Hs)."""
    n = s.size
    d = np.zeros((n, n))
    sy = s @ y
    if abs(sy) >= DENOM:
        d = d + np.outer(y, y) / sy
    Hs = H @ s
    sHs = s @ Hs
    if abs(sHs) >= DENOM:
        d = d - np.outer(Hs, Hs) / sHs
    return d


def sr1_delta(A, s):
    """hessian_update.py:67-85."""
    den = A @ s
    if abs(den) >= DENOM:
        return np.outer(A, A) / den
    return np.zeros((s.size, s.size))


def psb_delta(H, s, y):
    """hessian_update.py:87-104."""
    A = y - H @ s
    ss = s @ s
    if abs(ss) >= DENOM:
        return -(A @ s) * np.outer(s, s) / ss ** 2 + (np.outer(A, s) + np.outer(s, A)) / ss
    return np.zeros((s.size, s.size))


def bofill_const(A, s):
    """hessian_update.py:106-130: phi^2 = (A.s)^2 / ((A.A)(s.s))."""
    num = (A @ s) * (A @ s)
    den = (A @ A) * (s @ s)
    if abs(den) >= DENOM:
        return num / den
    return 0.0


def double_damping(s, y, mu2=0.2, thr=DENOM):
    """hessian_update.py:200-242 / block_hessian_update.py:565-595 (Powell damping, B = I)."""
    sy = s @ y
    ss = s @ s
    if sy < mu2 * ss:
        den = ss - sy
        theta = 0.1 if abs(den) < thr else (1.0 - mu2) * ss / den
        theta = min(max(theta, 0.0), 1.0)
        y = theta * y + (1.0 - theta) * s
    return s, y


def fsb_delta(H, s, y, cfd=False):
    """hessian_update.py:261-284."""
    A = y - H @ s
    if cfd:
        A = 2.0 * A
    phi = np.sqrt(bofill_const(A, s))
    return (1.0 - phi) * bfgs_delta(H, s, y) + phi * sr1_delta(A, s)


def bofill_delta(H, s, y, cfd=False):
    """hessian_update.py:286-307."""
    A = y - H @ s
    if cfd:
        A = 2.0 * A
    c = bofill_const(A, s)
    return (1.0 - c) * psb_delta(H, s, y) + c * sr1_delta(A, s)


def pcfd_bofill_delta(H, s, y):
    """hessian_update.py:309-343.

    The double loop over an orthonormal basis O of the complement of s sums
    (o_j^T D o_i)(o_i o_j^T + o_j o_i^T) = O(O^T D O)O^T + transpose = 2 P D P,
    P = O O^T = I - s s^T/(s^T s)  (D symmetric), independent of the basis chosen
    by scipy.linalg.null_space.
    """
    D = bofill_delta(H, s, y, cfd=True)
    P = np.eye(s.size) - np.outer(s, s) / (s @ s)
    PDP = P @ D @ P
    return D + PDP + PDP.T


def msp_delta(H, s, y):
    """hessian_update.py:345-368."""
    A = y - H @ s
    den = np.linalg.norm(A) * np.linalg.norm(s)
    c = 0.0
    if den >= DENOM:
        c = min(max((s @ A) / den, -1.0), 1.0)
    phi = 1.0 - c * c
    return phi * psb_delta(H, s, y) + (1.0 - phi) * sr1_delta(A, s)


def flowchart_delta(H, s, y):
    """hessian_update.py:163-194 (note z = y - H y, as written there)."""
    z = y - H @ y
    zs_den = np.linalg.norm(s) * np.linalg.norm(z)
    if abs(zs_den) < DENOM:
        zs_den += DENOM
    zs = (z @ s) / zs_den
    ys_den = np.linalg.norm(s) * np.linalg.norm(y)
    if abs(ys_den) < DENOM:
        ys_den += DENOM
    ys = (y @ s) / ys_den
    if zs < -0.1:
        return sr1_delta(y - H @ s, s)
    if ys > 0.1:
        return bfgs_delta(H, s, y)
    return fsb_delta(H, s, y)


# ----------------------------------------------------------------------------
# block rules at q = 1 (block_hessian_update.py)
# ----------------------------------------------------------------------------
def _symm(A):
    return 0.5 * (A + A.T)


def _safe_inv1(m):
    """safe_inv on a 1x1 matrix: LAPACK only reports singular for an exact zero pivot."""
    if m == 0.0:
        return 1.0 / (m + INV_REG)
    return 1.0 / m


def _blk_bfgs(B, s, y, damp=False):
    """block_hessian_update.py:75-118 (and :619-641 for the DD form)."""
    if not np.linalg.norm(s) > 1e-8:
        return B.copy()
    if damp:
        s, y = double_damping(s, y, 0.2, BLK_DENOM)
    elif (y @ s) <= BLK_DENOM:
        return B.copy()
    Bs = B @ s
    sB = B.T @ s
    m1 = s @ Bs
    m2 = s @ y
    Bp = B - np.outer(Bs, sB) * _safe_inv1(m1) + np.outer(y, y) * _safe_inv1(m2)
    return _symm(Bp)


def _blk_psb(B, s, y, thr=1e-8):
    """block_hessian_update.py:120-157."""
    if not np.linalg.norm(s) > thr:
        return B.copy()
    A = y - B @ s
    ss = float(s @ s)
    if abs(ss) >= thr:
        return B + (-(A @ s) * np.outer(s, s) / ss ** 2 + (np.outer(A, s) + np.outer(s, A)) / ss)
    return B + np.zeros_like(B)


def _blk_sr1(B, s, y, cfd=False):
    """block_hessian_update.py:159-184."""
    R = y - B @ s
    if cfd:
        R = 2.0 * R
    m = s @ R
    return _symm(B + np.outer(R, R) * _safe_inv1(m))


def _blk_weight(B, s, y, cfd, bofill_logic):
    """block_hessian_update.py:190-231."""
    A = y - B @ s
    if cfd:
        A = 2.0 * A
    num = (A @ s) ** 2
    den = (A @ A) * (s @ s)
    c = num / den if abs(den) > BLK_DENOM else 0.0
    if np.isnan(c):
        c = 0.0
    c = float(max(0.0, min(1.0, c)))
    return c if bofill_logic else np.sqrt(c)


def _blk_mix(B, s, y, kind, cfd, damp=False):
    """Mean-weight mixes, block_hessian_update.py:237-313 and :643-669."""
    if damp:
        s, y = double_damping(s, y, 0.2, BLK_DENOM)
    bofill_logic = (kind == "bofill") or cfd
    w = _blk_weight(B, s, y, cfd, bofill_logic)
    d_sr1 = _blk_sr1(B, s, y, cfd) - B
    if kind == "fsb":
        d_other = _blk_bfgs(B, s, y) - B
    else:
        d_other = _blk_psb(B, s, y) - B
    return _symm(B + w * d_sr1 + (1.0 - w) * d_other)


def _blk_mix_weighted(B, s, y, kind, cfd):
    """Weighted-subspace mixes, block_hessian_update.py:319-437 (q = 1: W = [w])."""
    bofill_logic = (kind == "bofill") or cfd
    w = _blk_weight(B, s, y, cfd, bofill_logic)
    d_sr1 = _blk_sr1(B, w * s, w * y, cfd) - B
    if kind == "fsb":
        d_other = _blk_bfgs(B, (1.0 - w) * s, (1.0 - w) * y) - B
    else:
        d_other = _blk_psb(B, (1.0 - w) * s, (1.0 - w) * y) - B
    return _symm(B + d_sr1 + d_other)


# ----------------------------------------------------------------------------
# dispatch (rsirfo.py:208-251; first substring match wins, else flowchart)
# ----------------------------------------------------------------------------
def _std(fn, **kw):
    return lambda H, s, y: fn(H, s, y, **kw)


def _dd(fn, **kw):
    def f(H, s, y):
        s2, y2 = double_damping(s, y, 0.2, DENOM)
        return fn(H, s2, y2, **kw)
    return f


DISPATCH = [
    ("flowchart", flowchart_delta),
    ("block_cfd_fsb_dd", lambda B, s, y: _blk_mix(B, s, y, "fsb", True, damp=True) - B),
    ("block_cfd_fsb_weighted", lambda B, s, y: _blk_mix_weighted(B, s, y, "fsb", True) - B),
    ("block_cfd_fsb", lambda B, s, y: _blk_mix(B, s, y, "fsb", True) - B),
    ("block_cfd_bofill_weighted", lambda B, s, y: _blk_mix_weighted(B, s, y, "bofill", True) - B),
    ("block_cfd_bofill", lambda B, s, y: _blk_mix(B, s, y, "bofill", True) - B),
    ("block_bfgs_dd", lambda B, s, y: _blk_bfgs(B, s, y, damp=True) - B),
    ("block_bfgs", lambda B, s, y: _blk_bfgs(B, s, y) - B),
    ("block_fsb_dd", lambda B, s, y: _blk_mix(B, s, y, "fsb", False, damp=True) - B),
    ("block_fsb_weighted", lambda B, s, y: _blk_mix_weighted(B, s, y, "fsb", False) - B),
    ("block_fsb", lambda B, s, y: _blk_mix(B, s, y, "fsb", False) - B),
    ("block_bofill_weighted", lambda B, s, y: _blk_mix_weighted(B, s, y, "bofill", False) - B),
    ("block_bofill", lambda B, s, y: _blk_mix(B, s, y, "bofill", False) - B),
    ("bfgs_dd", _dd(bfgs_delta)),
    ("bfgs", bfgs_delta),
    ("sr1", lambda H, s, y: sr1_delta(y - H @ s, s)),
    ("pcfd_bofill", pcfd_bofill_delta),
    ("cfd_fsb_dd", _dd(fsb_delta, cfd=True)),
    ("cfd_fsb", _std(fsb_delta, cfd=True)),
    ("cfd_bofill", _std(bofill_delta, cfd=True)),
    ("fsb_dd", _dd(fsb_delta)),
    ("fsb", fsb_delta),
    ("bofill", bofill_delta),
    ("psb", psb_delta),
    ("msp", msp_delta),
]


def select_rule(method):
    """Return (key, fn) exactly as the reference's priority scan does."""
    low = method.lower()
    for key, fn in DISPATCH:
        if key in low:
            return key, fn
    return "flowchart", flowchart_delta


def delta_hessian(method, H, s, y):
    key, fn = select_rule(method)
    return fn(H, np.asarray(s, dtype=float).ravel(), np.asarray(y, dtype=float).ravel())
