"""Early stop of the persistent tridiagonalisation on a diagonal trailing block (k_tri_persist3, round 3) and the skipping of trivial
reflector blocks behind it (k_tri_larft, k_tri_backtrans_mfma, k_apply_qt): eigen-solver properties against LAPACK on matrices that
deflate early, late, immediately, never, and on matrices built to fool the test (a negligible pivot-row tail in front of a dense block)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(A, tol_w=1e-13, tol_o=1e-11, tol_r=5e-12):
    from multioptpy_amd import ops
    import test_gpu_configs as t
    return t._check_eigh(ops, A, tol_w, tol_o, tol_r)


def _low_rank(n, r, rng, scale=1.0, diag=None):
    U = rng.normal(size=(n, r))
    S = rng.normal(size=(r, r)); S = 0.5 * (S + S.T) * scale
    D = np.eye(n) if diag is None else np.diag(diag)
    A = D + U @ S @ U.T
    return 0.5 * (A + A.T)


@pytest.mark.parametrize("n,batch", [(300, 32), (120, 8), (600, 16), (600, 64), (450, 4), (339, 5)])
def test_identity_plus_low_rank(n, batch):
    """I + U S U^T (ranks 3 .. 40): the Krylov space of e_1 is exhausted after rank + 1 steps, the trailing block is the identity."""
    rng = np.random.default_rng(n + batch)
    A = np.stack([_low_rank(n, 3 + (37 * b) // max(1, batch - 1), rng, scale=0.3 + b % 3) for b in range(batch)])
    _check(A)


def test_low_rank_on_a_non_constant_diagonal_does_not_deflate_wrongly():
    """diag(d) + low rank with distinct d: the trailing block is NOT diagonal after the reflectors (they mix the d_i): full reduction."""
    rng = np.random.default_rng(3)
    n = 200
    A = np.stack([_low_rank(n, 5, rng, diag=np.linspace(1.0, 2.0, n)) for _ in range(4)])
    _check(A)


def test_diagonal_and_block_diagonal_inputs():
    """A diagonal matrix stops at the first step; diag(B, I) with a dense leading block stops behind it; diag(I, B): the first pivot rows
    have zero tails while the trailing block is dense -- the test must say no and the reduction must go on."""
    rng = np.random.default_rng(9)
    n = 256
    B = rng.normal(size=(96, 96)); B = 0.5 * (B + B.T)
    A0 = np.diag(rng.normal(size=n))
    A1 = np.eye(n); A1[:96, :96] = B
    A2 = np.eye(n); A2[-96:, -96:] = B
    A3 = np.eye(n) * 2.5
    _check(np.stack([A0, A1, A2, A3]))


def test_noise_level_coupling_is_dropped_not_amplified():
    """Identity + low rank + symmetric noise of 1e-17: below the deflation tolerance (8 eps ||T||), so the reduction stops early; the
    eigenvalues must still agree with LAPACK to 1e-13 ||A||.  And noise of 1e-12 -- above the tolerance -- must be resolved (no stop)."""
    rng = np.random.default_rng(21)
    n = 300
    for lvl in (1e-17, 1e-12):
        N = rng.normal(size=(n, n)) * lvl
        A = np.stack([_low_rank(n, 12, rng) + 0.5 * (N + N.T) for _ in range(8)])
        _check(A)


def test_band_long_run_identity_start_vs_oracle():
    """Twelve band steps from identity Hessians (8 x 40): the regime in which every solve stops early, against the oracle."""
    import test_gpu_configs as t
    worst, _ = t._run_band(8, 40, 12, False)
    print("8 x 40 band, 12 steps from identity: worst deviation", worst)
