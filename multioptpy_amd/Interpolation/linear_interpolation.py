"""Drop-in for multioptpy/Interpolation/linear_interpolation.py::distribute_geometry (nebmain's `-ad N`, neb.py:649-766): images at
equal arc-length intervals along the piecewise linear path.  Host glue on (I, N, 3) arrays, like the reference's; the band's
geometry is on the host at this point of an iteration anyway (it is what the calculator is called with next)."""
import numpy as np

from ..Utils.calc_tools import calc_path_length_list


def distribute_geometry(geometry_list):
    """linear_interpolation.py:308-336.  The node at arc length d goes into the FIRST segment j with L[j] <= d <= L[j+1] (so a node that
    falls on a joint is taken from the earlier segment); a d that no segment brackets (rounding at the far end) becomes the last image."""
    geoms = [np.asarray(g, dtype="float64") for g in geometry_list]
    nnode = len(geoms)
    L = np.asarray(calc_path_length_list(geoms), dtype="float64")
    total = L[-1]
    if total < 1e-8:
        return list(geometry_list)
    node_dist = total / (nnode - 1)
    out = [geoms[0]]
    lo, hi = L[:-1], L[1:]
    for i in range(1, nnode - 1):
        d = i * node_dist
        hit = np.nonzero((lo <= d) & (d <= hi))[0]
        if hit.size == 0:
            out.append(geoms[-1])
            continue
        j = int(hit[0])
        t = (d - L[j]) / (L[j + 1] - L[j])
        out.append(geoms[j] + (geoms[j + 1] - geoms[j]) * t)
    out.append(geoms[-1])
    return out
