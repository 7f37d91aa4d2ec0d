"""ORACLE (test infrastructure, never imported by the product): the harmonic distance restraint of the reference
(multioptpy/Potential/keep_potential.py:21-58, StructKeepPotential.calc_energy) with the gradient and Hessian the reference obtains by
automatic differentiation (Potential/potential.py:127-152: torch.func.jacrev / torch.func.hessian), written out analytically.
  E = 1/2 k (r - r0)^2,  r = max(|x_i - x_j|, 1e-12),  r0 given in angstrom
Pinned by tests/golden/g14_keep_pot.npz (outputs of the reference)."""
import numpy as np

BOHR2ANG = 0.52917721067


def keep_pot(x, pair_1based, k, r0_ang):
    """x (N,3) bohr -> (E, grad (N,3), hess (3N,3N))."""
    x = np.asarray(x, dtype=float).reshape(-1, 3)
    N = len(x)
    i, j = pair_1based[0] - 1, pair_1based[1] - 1
    r0 = r0_ang / BOHR2ANG
    d = x[i] - x[j]
    r = np.linalg.norm(d)
    g = np.zeros((N, 3))
    H = np.zeros((3 * N, 3 * N))
    if not r > 1e-12:                                   # keep_potential.py:53: clamped, constant
        return 0.5 * k * (1e-12 - r0) ** 2, g, H
    u = d / r
    g[i] += k * (r - r0) * u
    g[j] -= k * (r - r0) * u
    h = k * (np.outer(u, u) + (1.0 - r0 / r) * (np.eye(3) - np.outer(u, u)))
    H[3 * i:3 * i + 3, 3 * i:3 * i + 3] += h
    H[3 * j:3 * j + 3, 3 * j:3 * j + 3] += h
    H[3 * i:3 * i + 3, 3 * j:3 * j + 3] -= h
    H[3 * j:3 * j + 3, 3 * i:3 * i + 3] -= h
    return 0.5 * k * (r - r0) ** 2, g, H


def anharmonic_keep_pot(x, pair_1based, k, D, r0_ang):
    """StructAnharmonicKeepPotential.calc_energy (anharmonic_keep_potential.py:13-27) with its derivatives: E = D (1 - exp(-a (r - r0)))^2,
    a = sqrt(k / 2D); zero for D = 0."""
    x = np.asarray(x, dtype=float).reshape(-1, 3)
    N = len(x)
    g = np.zeros((N, 3))
    H = np.zeros((3 * N, 3 * N))
    if D == 0.0:
        return 0.0, g, H
    i, j = pair_1based[0] - 1, pair_1based[1] - 1
    d = x[i] - x[j]
    r = np.linalg.norm(d)
    u = d / r
    a = np.sqrt(k / (2.0 * D))
    ex = np.exp(-a * (r - r0_ang / BOHR2ANG))
    V, V1, V2 = D * (1 - ex) ** 2, 2 * D * a * ex * (1 - ex), 2 * D * a * a * ex * (2 * ex - 1)
    g[i] += V1 * u
    g[j] -= V1 * u
    h = V2 * np.outer(u, u) + (V1 / r) * (np.eye(3) - np.outer(u, u))
    H[3 * i:3 * i + 3, 3 * i:3 * i + 3] += h
    H[3 * j:3 * j + 3, 3 * j:3 * j + 3] += h
    H[3 * i:3 * i + 3, 3 * j:3 * j + 3] -= h
    H[3 * j:3 * j + 3, 3 * i:3 * i + 3] -= h
    return V, g, H
