"""StructKeepAnglePotential / StructKeepDihedralAnglePotential: device counterparts of
``multioptpy.Potential.keep_angle_potential.StructKeepAnglePotential`` (reference keep_angle_potential.py:7-224) and
``multioptpy.Potential.keep_dihedral_angle_potential.StructKeepDihedralAnglePotential`` (keep_dihedral_angle_potential.py:6-154).

The reference evaluates ``calc_energy`` with torch and differentiates it automatically on the CPU every optimizer step
(Potential/potential.py:127-152).  Here ``calc_energy_gradient_hessian`` returns energy, gradient and Hessian from ``mopt_keep_angle``
(csrc/k_biasang.hip: the same expressions, branch by branch, evaluated in hyper-dual arithmetic).  Constructor keywords, units (angles in
degrees, geometry in bohr, 1-based atom labels) and the meaning of ``bias_pot_params`` = [spring constant, angle in degrees] are the reference's."""
import ctypes as C

import numpy as np

from .. import _lib
from .._lib import lib, check, f64, ptr


def keep_angle_sum(geom_num_list, atoms_1based, spring_consts, angles_deg, dihedral=False, device=None, out_of_plane=False):
    """E, grad (N, 3), H (3N, 3N) of the sum of angle (three labels per term), dihedral-angle or out-of-plane-angle (four labels per term:
    for the latter centre, probe, two plane atoms) restraints."""
    X = f64(np.asarray(geom_num_list, dtype=np.float64)).reshape(-1, 3)
    N = X.shape[0]
    na = 4 if (dihedral or out_of_plane) else 3
    atoms = np.ascontiguousarray(np.asarray(atoms_1based, dtype=np.int32).reshape(-1, na) - 1)
    # the reference converts the equilibrium angle with torch.deg2rad on a float32 tensor (keep_angle_potential.py:71-73): single precision,
    # e.g. 180 degrees -> pi + 8.7e-8, which takes the general branch and not the `theta_0 == pi` one.  Reproduced.
    rad = (np.asarray(angles_deg, dtype=np.float32).reshape(-1) * np.float32(np.pi / 180.0)).astype(np.float64)
    kp = np.ascontiguousarray(np.stack([np.asarray(spring_consts, dtype=np.float64).reshape(-1), rad], axis=1))
    if len(kp) != len(atoms):
        raise ValueError("one spring constant and one angle per term")
    E = np.zeros(1)
    g = np.empty((N, 3))
    H = np.empty((3 * N, 3 * N))
    c = _lib.ctx(device)
    check(lib().mopt_keep_angle(c, 2 if out_of_plane else (1 if dihedral else 0), N, ptr(X), len(atoms), atoms.ctypes.data_as(C.POINTER(C.c_int)), ptr(kp), ptr(E), ptr(g), ptr(H)), c)
    return float(E[0]), g, H


class _AngleBase:
    _prefix = ""
    _dihedral = False
    _oop = False

    def __init__(self, **kwarg):
        self.config = kwarg

    def _params(self, bias_pot_params):
        p = np.asarray(bias_pot_params, dtype=float).reshape(-1) if bias_pot_params is not None else np.zeros(0)
        if len(p) == 0:
            return float(self.config[self._prefix + "_spring_const"]), float(self.config[self._prefix + "_angle"])
        return float(p[0]), float(p[1])

    def calc_energy_gradient_hessian(self, geom_num_list, bias_pot_params=(), device=None):
        k, ang = self._params(bias_pot_params)
        na = 4 if (self._dihedral or self._oop) else 3
        return keep_angle_sum(geom_num_list, [list(self.config[self._prefix + "_atom_pairs"])[:na]], [k], [ang], self._dihedral, device, out_of_plane=self._oop)

    def calc_energy(self, geom_num_list, bias_pot_params=()):
        return self.calc_energy_gradient_hessian(geom_num_list, bias_pot_params)[0]


class StructKeepAnglePotential(_AngleBase):
    """keep_angle_spring_const, keep_angle_angle (degrees), keep_angle_atom_pairs (three 1-based labels; the middle one is the apex)."""
    _prefix = "keep_angle"


class StructKeepDihedralAnglePotential(_AngleBase):
    """keep_dihedral_angle_spring_const, keep_dihedral_angle_angle (degrees), keep_dihedral_angle_atom_pairs (four 1-based labels)."""
    _prefix = "keep_dihedral_angle"
    _dihedral = True


class StructKeepOutofPlainAnglePotential(_AngleBase):
    """keep_out_of_plain_angle_spring_const, keep_out_of_plain_angle_angle (degrees), keep_out_of_plain_angle_atom_pairs (four 1-based labels:
    centre, probe, two plane atoms) -- reference keep_outofplain_angle_potential.py:6-146."""
    _prefix = "keep_out_of_plain_angle"
    _oop = True


class StructKeepPotentialv2:
    """Fragment-centre distance restraint, reference keep_potential.py:63-115: keep_pot_v2_spring_const, keep_pot_v2_distance (angstrom),
    keep_pot_v2_fragm1 / keep_pot_v2_fragm2 (1-based atom lists)."""

    def __init__(self, **kwarg):
        self.config = kwarg
        self.bohr2angstroms = 0.52917721067

    def calc_energy_gradient_hessian(self, geom_num_list, bias_pot_params=(), device=None):
        p = np.asarray(bias_pot_params, dtype=float).reshape(-1) if bias_pot_params is not None else np.zeros(0)
        k, r0 = (float(self.config["keep_pot_v2_spring_const"]), float(self.config["keep_pot_v2_distance"])) if len(p) == 0 else (float(p[0]), float(p[1]))
        X = f64(np.asarray(geom_num_list, dtype=np.float64)).reshape(-1, 3)
        N = X.shape[0]
        f1 = np.ascontiguousarray(np.asarray(self.config["keep_pot_v2_fragm1"], dtype=np.int32) - 1)
        f2 = np.ascontiguousarray(np.asarray(self.config["keep_pot_v2_fragm2"], dtype=np.int32) - 1)
        E = np.zeros(1)
        g = np.empty((N, 3))
        H = np.empty((3 * N, 3 * N))
        c = _lib.ctx(device)
        IP = C.POINTER(C.c_int)
        check(lib().mopt_keep_pot_v2(c, N, ptr(X), len(f1), f1.ctypes.data_as(IP), len(f2), f2.ctypes.data_as(IP), k, r0 / self.bohr2angstroms, ptr(E), ptr(g),
                                     ptr(H)), c)
        return float(E[0]), g, H

    def calc_energy(self, geom_num_list, bias_pot_params=()):
        return self.calc_energy_gradient_hessian(geom_num_list, bias_pot_params)[0]
