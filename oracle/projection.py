"""Oracle: translation/rotation projectors and Kabsch (test infrastructure, see oracle/__init__.py).

Restates
  multioptpy/Utils/calc_tools.py:249-316   project_out_hess_tr_and_rot_for_coord
  multioptpy/Optimizer/rsirfo.py:128-190    RSIRFO._project_grad_tr_rot
  multioptpy/Utils/calc_tools.py:412-425    kabsch_algorithm
"""
import numpy as np


def trrot_vectors(coords):
    """Six un-normalised translation / rotation vectors about the geometric centre.

    Row order: Tx, Ty, Tz, Rx=(0,-z,y), Ry=(z,0,-x), Rz=(-y,x,0)
    (calc_tools.py:266-288, rsirfo.py:141-167).  The centre is accumulated atom by
    atom as calc_tools.calc_center (:138-145) does.
    """
    coords = np.asarray(coords, dtype=float).reshape(-1, 3)
    N = coords.shape[0]
    centre = np.zeros(3)
    for a in range(N):
        centre += coords[a]
    centre /= N
    c = coords - centre
    B = np.zeros((6, 3 * N))
    for k in range(3):
        B[k, k::3] = 1.0
    B[3, 1::3] = -c[:, 2]
    B[3, 2::3] = c[:, 1]
    B[4, 0::3] = c[:, 2]
    B[4, 2::3] = -c[:, 0]
    B[5, 0::3] = -c[:, 1]
    B[5, 1::3] = c[:, 0]
    return B


def gram_schmidt_rows(B, tol=1e-10):
    """calc_tools.py:250-259: classical GS (projections of the ORIGINAL v), drop norm <= tol."""
    basis = []
    for v in B:
        w = v.copy()
        for b in basis:
            w -= (v @ b) * b
        nrm = np.linalg.norm(w)
        if nrm > tol:
            basis.append(w / nrm)
    return np.array(basis).reshape(len(basis), B.shape[1])


def project_hessian_trrot(H, coords):
    """P^T H P with P = I - sum v v^T, then (X + X^T)/2  (calc_tools.py:295-304)."""
    Q = gram_schmidt_rows(trrot_vectors(coords))
    n = H.shape[0]
    P = np.eye(n)
    for v in Q:
        P -= np.outer(v, v)
    Hp = P.T @ H @ P
    return (Hp + Hp.T) / 2


def householder_qr_q(A):
    """Reduced Q (n x k) of A (n x k) by LAPACK-convention Householder reflections.

    np.linalg.qr(mode='reduced') = dgeqrf + dorgqr.  For a rank-deficient A the
    surplus columns of Q are whatever the reflectors produce from e_j; this routine
    follows dlarfg (beta = -sign(alpha) * ||x||, H = I when the sub-column is zero).
    """
    A = np.array(A, dtype=float)
    n, k = A.shape
    vs, taus = [], []
    for j in range(k):
        x = A[j:, j].copy()
        alpha = x[0]
        xnorm = np.linalg.norm(x[1:]) if x.size > 1 else 0.0
        if xnorm == 0.0:
            tau = 0.0
            v = np.zeros_like(x)
            v[0] = 1.0
        else:
            beta = -np.copysign(np.hypot(alpha, xnorm), alpha)
            tau = (beta - alpha) / beta
            v = x / (alpha - beta)
            v[0] = 1.0
            A[j, j] = beta
        vs.append(v)
        taus.append(tau)
        if tau != 0.0 and j + 1 < k:
            w = v @ A[j:, j + 1:]
            A[j:, j + 1:] -= tau * np.outer(v, w)
    Q = np.zeros((n, k))
    Q[:k, :k] = np.eye(k)
    for j in reversed(range(k)):
        if taus[j] != 0.0:
            w = vs[j] @ Q[j:, :]
            Q[j:, :] -= taus[j] * np.outer(vs[j], w)
    return Q


def project_gradient_trrot(g, coords):
    """g - Q Q^T g with Q = reduced QR of the (n,6) TR/rot matrix (rsirfo.py:169-190)."""
    g = np.asarray(g, dtype=float).ravel()
    Q = householder_qr_q(trrot_vectors(coords).T)
    return g - Q @ (Q.T @ g)


def kabsch(P, Q):
    """calc_tools.py:412-425.  Centres P and Q IN PLACE (as the reference does),
    returns (rotated P, centred Q)."""
    cP = np.array([np.mean(P.T[0]), np.mean(P.T[1]), np.mean(P.T[2])])
    cQ = np.array([np.mean(Q.T[0]), np.mean(Q.T[1]), np.mean(Q.T[2])])
    P -= cP
    Q -= cQ
    Hm = P.T @ Q
    U, S, Vt = np.linalg.svd(Hm)
    R = Vt.T @ U.T
    if np.linalg.det(R) < 0:
        Vt[-1, :] *= -1
        R = Vt.T @ U.T
    return (R @ P.T).T, Q
