"""Drop-in for the members of multioptpy/Utils/calc_tools.py::Calculationtools that sit on the optimizer-step path."""
import numpy as np

from .. import ops


def calc_path_length_list(geometry_list):
    """calc_tools.py:853-862: cumulative distance between consecutive images, each image taken relative to its own centroid."""
    centred = [np.asarray(g, dtype="float64") - np.mean(np.asarray(g, dtype="float64"), axis=0) for g in geometry_list]
    out = [0.0]
    for a, b in zip(centred[:-1], centred[1:]):
        out.append(out[-1] + np.linalg.norm(b - a))
    return out


def apply_climbing_image(geometry_list, energy_list, element_list):
    """calc_tools.py:865-883 (nebmain's `-ci start interval`, neb.py:561-564): for every local maximum of the natural cubic spline of the
    energies over the path length that lies strictly inside a segment (i, i + 1) with 2 <= i <= I - 3, image i is replaced by the point
    of that segment at the maximum, relaxed on the image dependent pair potential between its two neighbours.  The list is modified
    in place and returned, as in the reference."""
    from ..Interpolation.interpolation import spline_interpolation
    from ..Potential.idpp import IDPP
    L = calc_path_length_list(geometry_list)
    maxima, _ = spline_interpolation(L, energy_list)
    for distance, _energy in maxima:
        for i in range(2, len(L) - 2):
            if L[i] >= distance or distance >= L[i + 1]:
                continue
            t = (distance - L[i]) / (L[i + 1] - L[i])
            mid = geometry_list[i] + (geometry_list[i + 1] - geometry_list[i]) * t
            trio = IDPP().opt_path([geometry_list[i], mid, geometry_list[i + 1]], element_list)
            geometry_list[i] = trio[1]
    return geometry_list


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
