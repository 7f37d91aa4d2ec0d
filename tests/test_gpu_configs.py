"""GPU parity at the BASELINE.json configuration sizes that round 1 never ran on the card (all through the C ABI):
  * nebmain 64 images x 200 atoms (n = 600): band steps vs the oracle, eigen-solver properties at n = 600;
  * nebmain 16 images x 14 atoms (the Claisen-sized band): identity and exact-Hessian start vs the oracle;
  * the 32 x 100 headline band started from exact LJ Hessians (the reference's `-fc 1` path,
    Calculator/lj_calculation_tools.py:340-353 -> neb.py:794-804): dense molecular spectra with the six-fold TR/rot zero
    cluster, vs the oracle.
Tolerance: 1e-8 (north_star) on geometries in angstrom; eigen-solver residuals relative to ||A||_inf."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _oracle_band_steps(X0, nsteps, hessians=None):
    from test_gpu_parity import _oracle_band_steps as f
    return f(X0, nsteps, literal=False, hessians=hessians)


def _run_band(I, N, nsteps, exact, tol=1e-8):
    from multioptpy_amd.band import Band
    X0 = cases.band(I, N)
    hess = [cases.lj(X0[i], hessian=True)[2] for i in range(I)] if exact else None
    band = Band(I, N)
    if exact:
        for i in range(I):
            band.set_hessian(i, hess[i])
    worst = 0.0
    for it, (X, E, Gr, ref) in enumerate(_oracle_band_steps(X0, nsteps, hessians=None if hess is None else [h.copy() for h in hess])):
        band.upload(X, E, Gr)
        band.step(it)
        newX, _ = band.download(want_force=True)
        err = np.abs(newX - ref).max()
        worst = max(worst, err)
        assert err < tol, (I, N, exact, it, err)
    Hd = band.get_hessian(I // 2)
    band.close()
    return worst, Hd


@pytest.mark.parametrize("exact", [False, True], ids=["identity", "exact"])
def test_band_16x14_vs_oracle(exact):
    """BASELINE config 4's shape (16-image band of a 14-atom system, n = 42): five consecutive steps."""
    worst, _ = _run_band(16, 14, 5, exact)
    print("16 x 14 band, exact Hessian start" if exact else "16 x 14 band, identity start", "worst deviation", worst)


def test_band_64x200_vs_oracle():
    """BASELINE config 5's shape (64 images x 200 atoms, n = 600): two band steps against the oracle (closed-form BNEB
    projection; the oracle's literal SVD form needs minutes at this size and is pinned at N <= 14 by G6)."""
    worst, Hd = _run_band(64, 200, 2, False)
    assert np.abs(Hd - Hd.T).max() < 1e-12
    print("64 x 200 band: worst deviation from the oracle", worst, "angstrom")


def test_band_headline_exact_hessians_vs_oracle():
    """32 x 100 from exact LJ Hessians: three steps against the oracle."""
    worst, _ = _run_band(32, 100, 3, True)
    print("32 x 100 band from exact Hessians: worst deviation", worst)


def _check_eigh(ops, A, tol_w=1e-13, tol_o=1e-11, tol_r=5e-12):
    w, v = ops.eigh(A)
    if A.ndim == 2:
        w, v, A = w[None], v[None], A[None]
    for b in range(A.shape[0]):
        n = A.shape[1]
        wr = np.linalg.eigvalsh(A[b])
        nrm = np.abs(A[b]).sum(axis=1).max()
        assert np.all(np.diff(w[b]) >= 0)
        assert np.abs(w[b] - wr).max() < tol_w * nrm, (b, np.abs(w[b] - wr).max(), nrm)
        assert np.abs(v[b].T @ v[b] - np.eye(n)).max() < tol_o, (b, np.abs(v[b].T @ v[b] - np.eye(n)).max())
        assert np.abs(A[b] @ v[b] - v[b] * w[b]).max() < tol_r * nrm, (b, np.abs(A[b] @ v[b] - v[b] * w[b]).max())
    return w, v


def test_eigh_n600_properties():
    """n = 600: a random symmetric matrix, the TR/rot-projected LJ Hessian of a 200-atom cluster (six-fold zero cluster inside a
    dense spectrum), the projected identity (594-fold eigenvalue), one batch each and all three in one batch."""
    from multioptpy_amd import ops
    from oracle import projection as pj
    n = 600
    rng = np.random.default_rng(600)
    R = rng.normal(size=(n, n))
    R = 0.5 * (R + R.T)
    x = cases.cluster(200)
    H = pj.project_hessian_trrot(cases.lj(x, hessian=True)[2], x)
    P1 = pj.project_hessian_trrot(np.eye(n), x)
    w, _ = _check_eigh(ops, np.array([R, H, P1]))
    assert np.sum(np.abs(w[1]) < 1e-9 * np.abs(w[1]).max()) == 6
    assert np.sum(np.abs(w[2]) < 1e-10) == 6 and np.sum(np.abs(w[2] - 1.0) < 1e-10) == 594
    _check_eigh(ops, R)


def test_eigh_batch64_n600():
    """The 64 x 600 batch of BASELINE config 5 in one call (several persistent launches): every matrix correct."""
    from multioptpy_amd import ops
    n, B = 600, 64
    rng = np.random.default_rng(64600)
    base = rng.normal(size=(n, n))
    base = 0.5 * (base + base.T)
    A = np.empty((B, n, n))
    for b in range(B):
        u = rng.normal(size=(n, 3))
        A[b] = base + (b + 1) * 0.01 * (u @ u.T)
    w, v = ops.eigh(A)
    for b in (0, 7, 8, 31, 32, 63):
        wr = np.linalg.eigvalsh(A[b])
        nrm = np.abs(A[b]).sum(axis=1).max()
        assert np.abs(w[b] - wr).max() < 1e-13 * nrm
        assert np.abs(v[b].T @ v[b] - np.eye(n)).max() < 1e-11
        assert np.abs(A[b] @ v[b] - v[b] * w[b]).max() < 5e-12 * nrm


def test_rsprfo_bofill_n66_trace():
    """G3 p06: EnhancedRSPRFO rsprfo_bofill, saddle order 1, N = 22 (BASELINE config 3's size) against the reference."""
    import os
    import test_oracle_golden as tog
    from multioptpy_amd.Optimizer.rsprfo import EnhancedRSPRFO
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(cases.__file__)), "g3_rsprfo.npz"))
    tag = "p06"
    assert int(z[tag + "_meta"][0]) == 22 and str(z[tag + "_method"]) == "rsprfo_bofill"
    for k, opt, step in tog._replay_rsprfo(z, tag, EnhancedRSPRFO):
        ref = z[f"{tag}_s{k}_step"]
        assert np.abs(step - ref).max() < 1e-8 * max(1.0, np.abs(ref).max()), k
        assert abs(opt.trust_radius - float(z[f"{tag}_s{k}_tr"])) < 1e-12
        assert np.abs(opt.hessian - z[f"{tag}_s{k}_H"]).max() < 1e-8
        assert int(opt.ts_mode_idx) == int(z[f"{tag}_s{k}_tsidx"])
