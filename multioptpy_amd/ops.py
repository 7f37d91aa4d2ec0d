"""Thin NumPy-facing wrappers over the building-block entry points of libmopt_hip.so."""
import ctypes as C

import numpy as np

from . import _abi
from ._lib import lib, ctx, check, f64, ptr, IP


def hess_update(method, H, s, y, want_inplace=True):
    """delta-H application for a batch: H (B,n,n) or (n,n); returns (H_inplace, H_sym, info)."""
    H = f64(H)
    single = H.ndim == 2
    H3 = H.reshape((-1,) + H.shape[-2:])
    B, n = H3.shape[0], H3.shape[1]
    s = f64(s).reshape(B, n)
    y = f64(y).reshape(B, n)
    Hi = np.empty_like(H3) if want_inplace else None
    Hs = np.empty_like(H3)
    info = np.zeros((B, 8))
    c = ctx()
    check(lib().mopt_hess_update(c, _abi.rule_from_method(method), n, B, ptr(H3), ptr(s), ptr(y), ptr(Hi), ptr(Hs), ptr(info)), c)
    if single:
        return (Hi[0] if Hi is not None else None), Hs[0], info[0]
    return Hi, Hs, info


def project_hess(H, X):
    H = f64(H)
    H3 = H.reshape((-1,) + H.shape[-2:])
    B, n = H3.shape[0], H3.shape[1]
    X = f64(X).reshape(B, n)
    out = np.empty_like(H3)
    c = ctx()
    check(lib().mopt_project_trrot_hess(c, n, B, ptr(H3), ptr(X), ptr(out)), c)
    return out.reshape(H.shape)


def project_grad(g, X):
    g = f64(g)
    n = g.shape[-1] if g.ndim > 1 else g.size
    g2 = g.reshape(-1, n)
    B = g2.shape[0]
    X = f64(X).reshape(B, n)
    out = np.empty_like(g2)
    c = ctx()
    check(lib().mopt_project_trrot_grad(c, n, B, ptr(g2), ptr(X), ptr(out)), c)
    return out.reshape(g.shape)


def eigh(A, return_sweeps=False):
    A = f64(A)
    A3 = A.reshape((-1,) + A.shape[-2:])
    B, n = A3.shape[0], A3.shape[1]
    W = np.empty((B, n))
    V = np.empty_like(A3)
    sw = np.zeros(B, dtype=np.int32)
    c = ctx()
    check(lib().mopt_eigh(c, n, B, ptr(A3), ptr(W), ptr(V), sw.ctypes.data_as(IP)), c)
    if A.ndim == 2:
        W, V = W[0], V[0]
    return (W, V, sw) if return_sweeps else (W, V)


def bneb_force(X, E, G, optimize_num=0, apply_ci_neb=99999):
    X = f64(X)
    I, N = X.shape[0], X.shape[1]
    E = f64(E)
    G = f64(G)
    F = np.empty_like(X)
    tau = np.empty_like(X)
    c = ctx()
    check(lib().mopt_bneb_force(c, I, N, ptr(X), ptr(E), ptr(G), int(optimize_num), int(apply_ci_neb), ptr(F), ptr(tau)), c)
    return F, tau


def dgemm(A, B):
    A = f64(A)
    B = f64(B)
    A3 = A.reshape((-1,) + A.shape[-2:])
    B3 = B.reshape((-1,) + B.shape[-2:])
    out = np.empty_like(A3)
    c = ctx()
    check(lib().mopt_dgemm(c, A3.shape[1], A3.shape[0], ptr(A3), ptr(B3), ptr(out)), c)
    return out.reshape(A.shape)


NEB_VARIANTS = {"bneb": 0, "bneb3": 1, "ewbneb": 2, "dneb": 3}


def neb_force(variant, X, E, G, optimize_num=0, apply_ci_neb=99999):
    """calc_force of the NEB force classes; returns (F, tau) for "bneb", F otherwise."""
    v = NEB_VARIANTS[variant]
    if v == 0:
        return bneb_force(X, E, G, optimize_num, apply_ci_neb)
    X = f64(X)
    I, N = X.shape[0], X.shape[1]
    E = f64(E)
    G = f64(G)
    F = np.empty_like(X)
    c = ctx()
    check(lib().mopt_neb_force(c, v, I, N, ptr(X), ptr(E), ptr(G), int(optimize_num), int(apply_ci_neb), ptr(F), None), c)
    return F


def quadratic_model(H, g, s):
    """g.s + 1/2 s^T H s (TrustRadius.update_trust_radii, trust_radius.py:146-151); H (n,n) or (B,n,n)."""
    H = f64(H)
    H3 = H.reshape((-1,) + H.shape[-2:])
    B, n = H3.shape[0], H3.shape[1]
    g = f64(g).reshape(B, n)
    s = f64(s).reshape(B, n)
    out = np.zeros(B)
    c = ctx()
    check(lib().mopt_quadratic_model(c, n, B, ptr(H3), ptr(g), ptr(s), ptr(out)), c)
    return float(out[0]) if H.ndim == 2 else out


def clip_and_move(x_bohr, step, trust_radius, bohr2ang):
    """optimizer.py:789-812: returns (new_geometry [angstrom], move_vector [bohr], ||step|| before clipping), flat arrays."""
    x = f64(x_bohr).reshape(-1)
    st = f64(step).reshape(-1)
    new = np.empty_like(x)
    mv = np.empty_like(x)
    nrm = np.zeros(1)
    c = ctx()
    check(lib().mopt_clip_and_move(c, x.size, ptr(x), ptr(st), float(trust_radius), float(bohr2ang), ptr(new), ptr(mv), ptr(nrm)), c)
    return new, mv, float(nrm[0])


def kabsch(P, Q):
    """calc_tools.py:412-425 without the in-place side effects: returns (rotated centred P, centred Q), both (N,3)."""
    P = f64(P).reshape(-1, 3)
    Q = f64(Q).reshape(-1, 3)
    Pa = np.empty_like(P)
    Qc = np.empty_like(Q)
    c = ctx()
    check(lib().mopt_kabsch(c, P.shape[0], ptr(P), ptr(Q), ptr(Pa), ptr(Qc)), c)
    return Pa, Qc
