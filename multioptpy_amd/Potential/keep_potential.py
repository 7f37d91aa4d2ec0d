"""StructKeepPotential: device counterpart of ``multioptpy.Potential.keep_potential.StructKeepPotential`` (reference
keep_potential.py:4-58): the harmonic distance restraint E = 1/2 k (r - r0)^2 between two atoms.

The reference evaluates ``calc_energy`` with torch and obtains the bias gradient and bias Hessian by automatic differentiation on
the CPU every optimizer step (Potential/potential.py:127-152).  Here ``calc_energy_gradient_hessian`` returns all three from the
analytic kernel of csrc/k_keep.hip (``mopt_keep_pot``); ``calc_energy`` keeps the reference's name and argument meaning (geometry
in bohr; ``bias_pot_params`` = [spring constant, distance in angstrom] overrides the constructor's ``keep_pot_spring_const`` /
``keep_pot_distance``; ``keep_pot_atom_pairs`` are 1-based atom labels).  ``keep_pot_sum`` evaluates several restraints in one call."""
import ctypes as C

import numpy as np

from .. import _lib
from .._lib import lib, check, f64, ptr

BOHR2ANGSTROMS = 0.52917721067


def keep_pot_sum(geom_num_list, pairs_1based, spring_consts, distances_ang, device=None, well_depths=None):
    """E, grad (N, 3), H (3N, 3N) of sum_p 1/2 k_p (r_p - r0_p)^2; pairs as 1-based labels, r0 in angstrom (the reference's units).
    ``well_depths`` (hartree, one per pair; None or a negative entry = harmonic) selects the anharmonic form D (1 - exp(-sqrt(k / 2D) (r - r0)))^2."""
    X = f64(np.asarray(geom_num_list, dtype=np.float64)).reshape(-1, 3)
    N = X.shape[0]
    pairs = np.ascontiguousarray(np.asarray(pairs_1based, dtype=np.int32).reshape(-1, 2) - 1)
    k = np.asarray(spring_consts, dtype=np.float64).reshape(-1)
    D = -np.ones_like(k) if well_depths is None else np.asarray(well_depths, dtype=np.float64).reshape(-1)
    kr = np.ascontiguousarray(np.stack([k, np.asarray(distances_ang, dtype=np.float64).reshape(-1) / BOHR2ANGSTROMS, D], axis=1))
    if len(kr) != len(pairs):
        raise ValueError("one spring constant and one distance per atom pair")
    E = np.zeros(1)
    g = np.empty((N, 3))
    H = np.empty((3 * N, 3 * N))
    c = _lib.ctx(device)
    check(lib().mopt_keep_pot(c, N, ptr(X), len(pairs), pairs.ctypes.data_as(C.POINTER(C.c_int)), ptr(kr), ptr(E), ptr(g), ptr(H)), c)
    return float(E[0]), g, H


class StructKeepPotential:
    def __init__(self, **kwarg):
        self.config = kwarg
        self.bohr2angstroms = BOHR2ANGSTROMS

    def _params(self, bias_pot_params):
        p = np.asarray(bias_pot_params, dtype=float).reshape(-1) if bias_pot_params is not None else np.zeros(0)
        if len(p) == 0:
            return float(self.config["keep_pot_spring_const"]), float(self.config["keep_pot_distance"])
        return float(p[0]), float(p[1])

    def calc_energy_gradient_hessian(self, geom_num_list, bias_pot_params=(), device=None):
        k, r0 = self._params(bias_pot_params)
        return keep_pot_sum(geom_num_list, [list(self.config["keep_pot_atom_pairs"])[:2]], [k], [r0], device)

    def calc_energy(self, geom_num_list, bias_pot_params=()):
        return self.calc_energy_gradient_hessian(geom_num_list, bias_pot_params)[0]


class StructAnharmonicKeepPotential:
    """Device counterpart of ``multioptpy.Potential.anharmonic_keep_potential.StructAnharmonicKeepPotential`` (reference
    anharmonic_keep_potential.py:5-27): E = D (1 - exp(-sqrt(k / 2D) (r - r0)))^2 between two atoms, zero for D = 0.  Constructor keywords
    as in the reference: anharmonic_keep_pot_spring_const, anharmonic_keep_pot_potential_well_depth, anharmonic_keep_pot_atom_pairs
    (1-based), anharmonic_keep_pot_distance (angstrom)."""

    def __init__(self, **kwarg):
        self.config = kwarg
        self.bohr2angstroms = BOHR2ANGSTROMS

    def calc_energy_gradient_hessian(self, geom_num_list, bias_pot_params=(), device=None):
        c = self.config
        return keep_pot_sum(geom_num_list, [list(c["anharmonic_keep_pot_atom_pairs"])[:2]], [float(c["anharmonic_keep_pot_spring_const"])],
                            [float(c["anharmonic_keep_pot_distance"])], device, well_depths=[float(c["anharmonic_keep_pot_potential_well_depth"])])

    def calc_energy(self, geom_num_list, bias_pot_params=()):
        return self.calc_energy_gradient_hessian(geom_num_list, bias_pot_params)[0]
