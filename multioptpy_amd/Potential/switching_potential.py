"""WellPotential, WellPotentialWall, WellPotentialVP, WellPotentialAround: device counterparts of the four well potentials of
``multioptpy.Potential.switching_potential`` (reference switching_potential.py:5-257).

One radial well -- a linear wall below ``a`` and beyond ``d``, quintic switches on [a, b] and [c, d), zero between -- on the distance of two
fragment centres, on |x|, |y| or |z| of the target atoms, on the distance of the target atoms from a fixed point, or from the centroid of a
set of centre atoms.  The reference evaluates ``calc_energy`` with torch and differentiates it automatically on the CPU every optimizer step
(Potential/potential.py:127-152); here ``calc_energy_gradient_hessian`` returns energy, gradient and Hessian from ``mopt_well_pot``
(csrc/k_biasang.hip: the same if / elif chain on the value of the distance, the selected expression differentiated in hyper-dual arithmetic).
Constructor keywords and units are the reference's: wall energy in kJ/mol, the four limit distances in angstrom, geometry in bohr, 1-based atom
labels; the point of WellPotentialVP is taken in single precision, as the reference's ``torch.tensor(..., dtype=torch.float32)`` does."""
import ctypes as C

import numpy as np

from .. import _lib
from .._lib import lib, check, f64, ptr

BOHR2ANG = 0.52917721067
HARTREE2KJMOL = 2625.500          # UnitValueLib().hartree2kjmol (Parameters/parameter.py)


def _well(kind, geom_num_list, list1, list2, wall_energy_kjmol, limit_dist_ang, point=(0.0, 0.0, 0.0), direction=0, device=None):
    X = f64(np.asarray(geom_num_list, dtype=np.float64)).reshape(-1, 3)
    N = X.shape[0]
    l1 = np.ascontiguousarray(np.asarray(list1, dtype=np.int32).reshape(-1) - 1)
    l2 = np.ascontiguousarray(np.asarray(list2, dtype=np.int32).reshape(-1) - 1)
    lim = [float(v) / BOHR2ANG for v in limit_dist_ang]
    pt = [float(np.float32(v)) for v in point]
    prm = np.array(lim + [float(wall_energy_kjmol) / HARTREE2KJMOL] + pt + [float(direction)], dtype=np.float64)
    E = np.zeros(1)
    g = np.empty((N, 3))
    H = np.empty((3 * N, 3 * N))
    c = _lib.ctx(device)
    IP = C.POINTER(C.c_int)
    check(lib().mopt_well_pot(c, int(kind), N, ptr(X), len(l1), l1.ctypes.data_as(IP), len(l2), l2.ctypes.data_as(IP) if len(l2) else None, ptr(prm), ptr(E), ptr(g),
                              ptr(H)), c)
    return float(E[0]), g, H


class _WellBase:
    def __init__(self, **kwarg):
        self.config = kwarg
        self.bohr2angstroms = BOHR2ANG
        self.hartree2kjmol = HARTREE2KJMOL

    def calc_energy(self, geom_num_list, bias_pot_params=()):
        return self.calc_energy_gradient_hessian(geom_num_list, bias_pot_params)[0]


class WellPotential(_WellBase):
    """well_pot_wall_energy, well_pot_fragm_1, well_pot_fragm_2, well_pot_limit_dist (switching_potential.py:5-67)"""

    def calc_energy_gradient_hessian(self, geom_num_list, bias_pot_params=(), device=None):
        cf = self.config
        return _well(0, geom_num_list, cf["well_pot_fragm_1"], cf["well_pot_fragm_2"], cf["well_pot_wall_energy"], cf["well_pot_limit_dist"], device=device)


class WellPotentialWall(_WellBase):
    """wall_well_pot_wall_energy, wall_well_pot_direction ("x" / "y" / "z"), wall_well_pot_limit_dist, wall_well_pot_target (switching_potential.py:70-135)"""

    def calc_energy_gradient_hessian(self, geom_num_list, bias_pot_params=(), device=None):
        cf = self.config
        direction = {"x": 0, "y": 1, "z": 2}[cf["wall_well_pot_direction"]]
        return _well(1, geom_num_list, cf["wall_well_pot_target"], [], cf["wall_well_pot_wall_energy"], cf["wall_well_pot_limit_dist"], direction=direction, device=device)


class WellPotentialVP(_WellBase):
    """void_point_well_pot_wall_energy, void_point_well_pot_coordinate (bohr), void_point_well_pot_limit_dist, void_point_well_pot_target
    (switching_potential.py:138-196)"""

    def calc_energy_gradient_hessian(self, geom_num_list, bias_pot_params=(), device=None):
        cf = self.config
        return _well(2, geom_num_list, cf["void_point_well_pot_target"], [], cf["void_point_well_pot_wall_energy"], cf["void_point_well_pot_limit_dist"],
                     point=np.asarray(cf["void_point_well_pot_coordinate"], dtype=float).reshape(3), device=device)


class WellPotentialAround(_WellBase):
    """around_well_pot_wall_energy, around_well_pot_center, around_well_pot_limit_dist, around_well_pot_target (switching_potential.py:199-257)"""

    def calc_energy_gradient_hessian(self, geom_num_list, bias_pot_params=(), device=None):
        cf = self.config
        return _well(3, geom_num_list, cf["around_well_pot_center"], cf["around_well_pot_target"], cf["around_well_pot_wall_energy"], cf["around_well_pot_limit_dist"], device=device)
