"""CaluculationDNEB: drop-in for ``multioptpy.MEP.pathopt_dneb_force.CaluculationDNEB`` (reference
pathopt_dneb_force.py:20-130; nebmain ``-dneb``), computed by k_neb_force_var on the GPU."""
import numpy as np

from .. import ops


class CaluculationDNEB:
    def __init__(self, APPLY_CI_NEB=99999):
        self.spring_constant_k = 0.01
        self.APPLY_CI_NEB = APPLY_CI_NEB
        self.force_const_for_cineb = 0.01

    def calc_force(self, geometry_num_list, energy_list, gradient_list, optimize_num, element_list):
        return ops.neb_force("dneb", np.asarray(geometry_num_list, dtype=np.float64), np.asarray(energy_list, dtype=np.float64),
                             np.asarray(gradient_list, dtype=np.float64), optimize_num, self.APPLY_CI_NEB)
