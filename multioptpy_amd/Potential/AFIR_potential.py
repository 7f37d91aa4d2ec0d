"""AFIRPotential: device counterpart of ``multioptpy.Potential.AFIR_potential.AFIRPotential`` (reference AFIR_potential.py:7-54).

The reference evaluates ``calc_energy`` with torch and obtains the bias gradient and bias Hessian by automatic differentiation on
the CPU every optimizer step (Potential/potential.py:127-152).  Here ``calc_energy_gradient_hessian`` returns all three from the
analytic kernels of csrc/k_afir.hip (``mopt_afir``); ``calc_energy`` keeps the reference's name and argument meaning
(geom in bohr, bias_pot_params[0] = gamma in kJ/mol).  Constructor keywords as in the reference: AFIR_Fragm_1, AFIR_Fragm_2
(1-based atom labels), element_list."""
import ctypes as C
import math

import numpy as np

from .. import _lib
from .._lib import lib, check, f64, ptr
from ..ModelHessian.approx_hessian import _tables

HARTREE2KJMOL = 2625.500
BOHR2ANGSTROMS = 0.52917721067


class AFIRPotential:
    def __init__(self, **kwarg):
        self.config = kwarg
        self.hartree2kjmol = HARTREE2KJMOL
        self.bohr2angstroms = BOHR2ANGSTROMS
        self.R_0 = 3.8164 / self.bohr2angstroms
        self.EPSIRON = 1.0061 / self.hartree2kjmol
        self.p = 6.0

    def alpha(self, gamma_kjmol):
        """AFIR_potential.py:34-37."""
        g = float(gamma_kjmol)
        if g == 0.0:
            return 0.0
        g /= self.hartree2kjmol
        return g / ((2 ** (-1 / 6) - (1 + math.sqrt(1 + abs(g) / self.EPSIRON)) ** (-1 / 6)) * self.R_0)

    def calc_energy_gradient_hessian(self, geom_num_list, bias_pot_params, device=None):
        X = f64(np.asarray(geom_num_list, dtype=np.float64)).reshape(-1, 3)
        N = X.shape[0]
        el = _tables()["elements"]
        cov = np.array([el[str(e)]["cov"] for e in self.config["element_list"]], dtype=np.float64)
        f1 = np.ascontiguousarray(np.asarray(self.config["AFIR_Fragm_1"], dtype=np.int32) - 1)
        f2 = np.ascontiguousarray(np.asarray(self.config["AFIR_Fragm_2"], dtype=np.int32) - 1)
        E = np.zeros(1)
        g = np.empty((N, 3))
        H = np.empty((3 * N, 3 * N))
        c = _lib.ctx(device)
        ip = C.POINTER(C.c_int)
        check(lib().mopt_afir(c, N, ptr(X), ptr(cov), len(f1), f1.ctypes.data_as(ip), len(f2), f2.ctypes.data_as(ip),
                              self.alpha(np.asarray(bias_pot_params, dtype=float).reshape(-1)[0]), ptr(E), ptr(g), ptr(H)), c)
        return float(E[0]), g, H

    def calc_energy(self, geom_num_list, bias_pot_params):
        return self.calc_energy_gradient_hessian(geom_num_list, bias_pot_params)[0]
