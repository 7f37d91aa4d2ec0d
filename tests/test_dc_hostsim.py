"""Divide-and-conquer tridiagonal eigen-solver: the scalar logic the kernels of k_dc.hip execute (multioptpy_amd/csrc/dc_logic.h),
built for the host with a serial reducer and the kernels' orchestration restated serially (tests/hostsim), against LAPACK.
No GPU needed."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import scipy.linalg as sla

import cases

HERE = os.path.dirname(os.path.abspath(__file__))
DP = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def sim():
    src = os.path.join(HERE, "hostsim", "hostsim.cpp")
    so = os.path.join(HERE, "hostsim", "libmopt_hostsim.so")
    hdrs = [os.path.join(HERE, "..", "multioptpy_amd", "csrc", f) for f in ("logic.h", "brent.h", "dc_logic.h")]
    if os.environ.get("MOPT_HOSTSIM_SO"):                    # (a sanitizer build handed in by tests/test_hostsim_sanitized.py)
        so = os.environ["MOPT_HOSTSIM_SO"]
    elif (not os.path.exists(so)) or any(os.path.getmtime(so) < os.path.getmtime(f) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-o", so, src])
    lib = C.CDLL(so)
    lib.sim_dc_eigh.argtypes = [C.c_int, DP, DP, DP, DP, C.POINTER(C.c_int)]
    return lib


def dc(sim, d, e):
    n = len(d)
    d = np.ascontiguousarray(d, dtype=float)
    e = np.ascontiguousarray(np.append(e, 0.0), dtype=float)
    W = np.zeros(n)
    Z = np.zeros((n, n))
    info = (C.c_int * 6)()
    sim.sim_dc_eigh(n, d.ctypes.data_as(DP), e.ctypes.data_as(DP), W.ctypes.data_as(DP), Z.ctypes.data_as(DP), info)
    return W, Z, list(info)


def check(sim, d, e, tag):
    n = len(d)
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    W, Z, info = dc(sim, d, e)
    nrm = max(np.abs(T).sum(axis=1).max(), 1e-300)
    wr = np.linalg.eigvalsh(T)
    o = np.argsort(W)
    assert info[0] == 0, (tag, "secular solves without convergence", info)
    assert info[5] == 0, (tag, "staged deflation (kernel layout) differs from the sequential scan", info)
    assert np.abs(W[o] - wr).max() <= 2e-14 * nrm * max(1.0, np.log2(n)), (tag, np.abs(W[o] - wr).max(), nrm)
    assert np.abs(Z @ Z.T - np.eye(n)).max() < 5e-13, (tag, "orthogonality", np.abs(Z @ Z.T - np.eye(n)).max())
    assert np.abs(T @ Z.T - Z.T * W).max() < 1e-13 * nrm * max(1.0, np.log2(n)), (tag, "residual", np.abs(T @ Z.T - Z.T * W).max() / nrm)
    return info


def tridiag_of(A):
    Hh = sla.hessenberg(A)
    return np.diag(Hh).copy(), np.diag(Hh, 1).copy()


def test_dc_random_and_sizes(sim):
    for n in (33, 64, 65, 100, 150, 257, 300, 450):
        rng = np.random.default_rng(n)
        check(sim, rng.normal(size=n), rng.normal(size=n - 1), f"random{n}")


def test_dc_dense_molecular_hessians(sim):
    """The regime this solver exists for: TR/rot-projected LJ Hessians (six-fold zero cluster inside a dense spectrum)."""
    from oracle import projection as pj
    for N in (22, 50, 100):
        x = cases.cluster(N)
        H = pj.project_hessian_trrot(cases.lj(x, hessian=True)[2], x)
        d, e = tridiag_of(H)
        info = check(sim, d, e, f"lj{N}")
        print("LJ", N, "info (fails, max iterations, rotations, deflated at the top)", info)


def test_dc_clusters_and_deflation(sim):
    n = 200
    # Wilkinson matrices (pairs of eigenvalues agreeing to many digits) and three of them glued with tiny couplings
    m = 61
    wd = np.abs(np.arange(m) - m // 2).astype(float)
    check(sim, wd, np.ones(m - 1), "wilkinson61")
    check(sim, np.concatenate([wd, wd, wd]), np.concatenate([np.ones(m - 1), [1e-9], np.ones(m - 1), [1e-7], np.ones(m - 1)]), "glued")
    # identity + low rank, reduced: massive deflation; a matrix with zero off-diagonals; constant diagonals (Toeplitz)
    rng = np.random.default_rng(5)
    U = rng.normal(size=(n, 4))
    d, e = tridiag_of(np.eye(n) + U @ U.T)
    check(sim, d, e, "identity+rank4")
    e0 = rng.normal(size=n - 1)
    e0[::7] = 0.0
    check(sim, rng.normal(size=n), e0, "split")
    check(sim, np.full(n, 2.0), np.full(n - 1, -1.0), "laplacian")
    check(sim, np.zeros(n), np.ones(n - 1), "zero-diagonal")
    check(sim, np.linspace(1, 2, n), 1e-12 * np.ones(n - 1), "nearly-diagonal")
    check(sim, 1e-8 * rng.normal(size=n), 1e-8 * rng.normal(size=n - 1), "tiny-scale")
    check(sim, 1e8 * rng.normal(size=n), 1e8 * rng.normal(size=n - 1), "large-scale")
    # graded matrix
    g = 10.0 ** np.linspace(0, -10, n)
    check(sim, g, 0.3 * np.sqrt(g[:-1] * g[1:]), "graded")
