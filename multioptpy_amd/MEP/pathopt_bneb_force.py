"""CaluculationBNEB: drop-in for ``multioptpy.MEP.pathopt_bneb_force.CaluculationBNEB``
(reference pathopt_bneb_force.py:21-222) -- the default NEB force of nebmain (neb.py:769-792).

``calc_force`` runs the per-atom closed form of the B-matrix projection on the GPU (k_bneb_force);
``tau_list`` / ``get_tau`` are kept because RFOOptimizer reads them (rfo_neb.py:54-68).
"""
import numpy as np

from .. import ops


class CaluculationBNEB:
    def __init__(self, APPLY_CI_NEB=99999):
        self.APPLY_CI_NEB = APPLY_CI_NEB
        self.tau_list = []

    def calc_ci_neb_force(self, grad, tangent_grad):
        return -2.0 * tangent_grad

    def calc_force(self, geometry_num_list, energy_list, gradient_list, optimize_num, element_list):
        X = np.asarray(geometry_num_list, dtype=np.float64)
        E = np.asarray(energy_list, dtype=np.float64)
        G = np.asarray(gradient_list, dtype=np.float64)
        F, tau = ops.bneb_force(X, E, G, optimize_num=optimize_num, apply_ci_neb=self.APPLY_CI_NEB)
        self.tau_list = [tau[i] for i in range(tau.shape[0])]
        return F

    def get_tau(self, node_num):
        if len(self.tau_list) == 0:
            raise ValueError("Tangent list is empty. Calculate forces first.")
        return self.tau_list[node_num]


class CaluculationBNEB3:
    """Drop-in for ``multioptpy.MEP.pathopt_bneb_force.CaluculationBNEB3`` (reference :305-356; nebmain ``-bneb2``)."""

    def __init__(self, APPLY_CI_NEB=99999):
        self.APPLY_CI_NEB = APPLY_CI_NEB
        self.spring_force_const = 0.05

    def calc_ci_neb_force(self, grad, tangent_grad):
        return -2.0 * tangent_grad

    def calc_force(self, geometry_num_list, energy_list, gradient_list, optimize_num, element_list):
        return ops.neb_force("bneb3", np.asarray(geometry_num_list, dtype=np.float64), np.asarray(energy_list, dtype=np.float64),
                             np.asarray(gradient_list, dtype=np.float64), optimize_num, self.APPLY_CI_NEB)
