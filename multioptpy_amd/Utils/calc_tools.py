"""Drop-in for the members of multioptpy/Utils/calc_tools.py::Calculationtools that sit on the optimizer-step path."""
import numpy as np

from .. import ops


class Calculationtools:
    def calc_center(self, geomerty, element_list=[]):
        """calc_tools.py:138-145 (plain centroid; the element list is ignored there too)."""
        return np.asarray(geomerty, dtype="float64").sum(axis=0) / len(geomerty)

    def project_out_hess_tr_and_rot_for_coord(self, hessian, element_list, geometry, display_eigval=False):
        """calc_tools.py:249-316 on the device (mopt_project_trrot_hess)."""
        return ops.project_hess(np.asarray(hessian, dtype="float64"), np.asarray(geometry, dtype="float64").reshape(-1))

    def kabsch_algorithm(self, P, Q):
        """calc_tools.py:412-425 (mopt_kabsch).  As in the reference both arguments are centred IN PLACE when they are
        float64 ndarrays (callers rely on it: optimization.py:1207-1213 shifts `state.geometry`, neb.py:993-996 shifts
        image k+1); returns (rotated P, centred Q) where the second item IS the caller's Q."""
        Pa, Qc = ops.kabsch(P, Q)
        if isinstance(P, np.ndarray) and P.dtype == np.float64:
            P -= np.array([np.mean(P.T[0]), np.mean(P.T[1]), np.mean(P.T[2])], dtype="float64")
        if isinstance(Q, np.ndarray) and Q.dtype == np.float64:
            Q[...] = Qc
            return Pa, Q
        return Pa, Qc
