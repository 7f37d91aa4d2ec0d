"""CaluculationEWBNEB: drop-in for ``multioptpy.MEP.pathopt_ewbneb_force.CaluculationEWBNEB`` (reference
pathopt_ewbneb_force.py:21-117; nebmain ``-ewbneb``), computed by k_neb_force_var on the GPU."""
import numpy as np

from .. import ops


class CaluculationEWBNEB:
    def __init__(self, APPLY_CI_NEB=99999):
        self.APPLY_CI_NEB = APPLY_CI_NEB
        self.upper_spring_constant = 0.005
        self.lower_spring_constant = 0.0001

    def calc_ci_neb_force(self, grad, tangent_grad):
        return -2.0 * tangent_grad

    def calc_force(self, geometry_num_list, energy_list, gradient_list, optimize_num, element_list):
        return ops.neb_force("ewbneb", np.asarray(geometry_num_list, dtype=np.float64), np.asarray(energy_list, dtype=np.float64),
                             np.asarray(gradient_list, dtype=np.float64), optimize_num, self.APPLY_CI_NEB)
