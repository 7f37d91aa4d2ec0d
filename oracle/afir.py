"""ORACLE (test infrastructure, never imported by the product): the AFIR bias potential of the reference
(multioptpy/Potential/AFIR_potential.py:8-54) with the gradient and Hessian the reference obtains by automatic differentiation
(Potential/potential.py:127-152: torch.func.jacrev / torch.func.hessian), written out analytically.
  E = alpha A / B,  A = sum_ij omega_ij r_ij,  B = sum_ij omega_ij,  omega_ij = ((R_i + R_j) / r_ij)^6  over i in fragment 1, j in fragment 2
Pinned by tests/golden/g11_afir.npz (outputs of the reference)."""
import math

import numpy as np

HARTREE2KJMOL = 2625.500
BOHR2ANG = 0.52917721067
R_0 = 3.8164 / BOHR2ANG
EPSIRON = 1.0061 / HARTREE2KJMOL


def afir_alpha(gamma_kjmol):
    if gamma_kjmol == 0.0:
        return 0.0
    g = gamma_kjmol / HARTREE2KJMOL
    return g / ((2 ** (-1 / 6) - (1 + math.sqrt(1 + abs(g) / EPSIRON)) ** (-1 / 6)) * R_0)


def afir(x, cov, frag1, frag2, gamma_kjmol):
    """x (N,3) bohr, cov (N,) covalent radii in bohr, frag1 / frag2 1-based atom labels -> (E, grad (N,3), hess (3N,3N))."""
    x = np.asarray(x, dtype=float).reshape(-1, 3)
    N = len(x)
    n = 3 * N
    alpha = afir_alpha(float(gamma_kjmol))
    A = B = 0.0
    gA, gB = np.zeros(n), np.zeros(n)
    HA, HB = np.zeros((n, n)), np.zeros((n, n))
    for i1 in frag1:
        for j1 in frag2:
            i, j = i1 - 1, j1 - 1
            d = x[i] - x[j]
            r = np.linalg.norm(d)
            u = d / r
            # the reference builds the radii as torch.tensor([...]) of default dtype: float32 values, added in float32
            c = float(np.float32(cov[i]) + np.float32(cov[j])) ** 6
            om, a = c / r ** 6, c / r ** 5
            A += a
            B += om
            for (val, d1, d2, gv, Hm) in ((a, -5 * c / r ** 6, 30 * c / r ** 7, gA, HA), (om, -6 * c / r ** 7, 42 * c / r ** 8, gB, HB)):
                gv[3 * i:3 * i + 3] += d1 * u
                gv[3 * j:3 * j + 3] -= d1 * u
                h = d2 * np.outer(u, u) + (d1 / r) * (np.eye(3) - np.outer(u, u))
                Hm[3 * i:3 * i + 3, 3 * i:3 * i + 3] += h
                Hm[3 * j:3 * j + 3, 3 * j:3 * j + 3] += h
                Hm[3 * i:3 * i + 3, 3 * j:3 * j + 3] -= h
                Hm[3 * j:3 * j + 3, 3 * i:3 * i + 3] -= h
    E = alpha * A / B
    g = alpha * (gA / B - A * gB / B ** 2)
    H = alpha * (HA / B - (np.outer(gA, gB) + np.outer(gB, gA)) / B ** 2 - A * HB / B ** 2 + 2 * A * np.outer(gB, gB) / B ** 3)
    return E, g.reshape(N, 3), H
