"""ORACLE (test infrastructure only -- never imported by the product): the four well potentials of the reference's
multioptpy/Potential/switching_potential.py (:5-257) with gradient and Hessian written out by hand.

The reference differentiates ``calc_energy`` with torch (Potential/potential.py:127-152).  Every term is a radial function w(r) of
r = |v|, v = sum_k s_k x_{a_k} (+ a constant point): dr/dx_a = s_a v / r, d2r/dx_a dx_b = s_a s_b (1 - v v^T / r^2) / r, so
    g_a = w' s_a v / r,      H_ab = s_a s_b [ w'' v v^T / r^2 + (w' / r)(1 - v v^T / r^2) ];
the wall variant takes r = |x_i[dir]| (dr = sign, no curvature).  w is chosen on the value of r as the reference's if / elif chain chooses it;
r = 0 (a target atom that is its own centre) contributes the value w(0) with zero derivatives, as torch.linalg.norm does."""
import numpy as np

BOHR2ANG = 0.52917721067
HARTREE2KJMOL = 2625.500


def well(r, lim_bohr, e0):
    """w, w', w'' at r (switching_potential.py:34-63)"""
    a, b, c, d = lim_bohr
    ss, si = 0.5 / (b - a), 1.0 - 0.5 * b / (b - a)
    ls, li = 0.5 / (c - d), 1.0 - 0.5 * c / (c - d)

    def quintic(x, slope):
        f = 2.0 - 20.0 * x ** 3 + 30.0 * x ** 4 - 12.0 * x ** 5
        f1 = (-60.0 * x ** 2 + 120.0 * x ** 3 - 60.0 * x ** 4) * slope
        f2 = (-120.0 * x + 360.0 * x ** 2 - 240.0 * x ** 3) * slope * slope
        return e0 * f, e0 * f1, e0 * f2
    if r <= a:
        return e0 * (-3.75 * (ss * r + si) + 2.875), e0 * (-3.75 * ss), 0.0
    if r <= b:
        return quintic(ss * r + si, ss)
    if r < c:
        return 0.0, 0.0, 0.0
    if r < d:
        return quintic(ls * r + li, ls)
    return e0 * (-3.75 * (ls * r + li) + 2.875), e0 * (-3.75 * ls), 0.0


def _radial_term(x, weights, const, lim, e0, g, H):
    """adds w(|sum_a s_a x_a - const|) to (E, g, H); weights: {atom: s}"""
    v = -np.asarray(const, dtype=float)
    for a, s in weights.items():
        v = v + s * x[a]
    r = float(np.linalg.norm(v))
    w, w1, w2 = well(r, lim, e0)
    if r == 0.0:
        return w
    u = v / r
    P = np.outer(u, u)
    blk = w2 * P + (w1 / r) * (np.eye(3) - P)
    for a, sa in weights.items():
        g[a] += w1 * sa * u
        for b, sb in weights.items():
            H[3 * a:3 * a + 3, 3 * b:3 * b + 3] += sa * sb * blk
    return w


def well_potential(kind, geom_bohr, list1, list2, wall_energy_kjmol, limit_dist_ang, point=(0.0, 0.0, 0.0), direction=0):
    """kind: "well" | "wall" | "vp" | "around"; 1-based atom labels; returns E, g (N, 3), H (3N, 3N)."""
    x = np.asarray(geom_bohr, dtype=float).reshape(-1, 3)
    N = len(x)
    lim = [float(v) / BOHR2ANG for v in limit_dist_ang]
    e0 = float(wall_energy_kjmol) / HARTREE2KJMOL
    g = np.zeros((N, 3))
    H = np.zeros((3 * N, 3 * N))
    E = 0.0
    # The summed kinds start from the Python float 0.0 and do `energy += term` (switching_potential.py:117-130): a term of the zero region is a
    # float32 tensor, the others are float64 tensors.  If the FIRST target lies in the zero region the accumulator becomes a float32 tensor and
    # every later in-place add is stored in single precision (computed in double, rounded on the way back; the value only: the derivatives pass
    # through the cast unchanged).
    state = {"first": True, "f32": False}
    a_, b_, c_, d_ = lim

    def acc(E, w, r):
        if state["first"]:
            state["first"] = False
            state["f32"] = (b_ < r < c_)
            return float(np.float32(w)) if state["f32"] else w
        return float(np.float32(E + w)) if state["f32"] else E + w      # (the add is carried out in the promoted type and stored as float32)

    def add(wts, a, s):
        wts[a] = wts.get(a, 0.0) + s
    if kind == "well":
        wts = {}
        for a in list1:
            add(wts, a - 1, 1.0 / len(list1))
        for a in list2:
            add(wts, a - 1, -1.0 / len(list2))
        E += _radial_term(x, wts, np.zeros(3), lim, e0, g, H)
    elif kind == "wall":
        for a in list1:
            comp = x[a - 1, direction]
            w, w1, w2 = well(abs(comp), lim, e0)
            E = acc(E, w, abs(comp))
            sg = 1.0 if comp > 0 else (-1.0 if comp < 0 else 0.0)
            g[a - 1, direction] += w1 * sg
            H[3 * (a - 1) + direction, 3 * (a - 1) + direction] += w2
    elif kind == "vp":
        pt = np.array([float(np.float32(v)) for v in point])          # torch.float32 (switching_potential.py:155)
        for a in list1:
            E = acc(E, _radial_term(x, {a - 1: 1.0}, pt, lim, e0, g, H), float(np.linalg.norm(x[a - 1] - pt)))
    elif kind == "around":
        for a in list2:
            wts = {}
            add(wts, a - 1, 1.0)
            for cidx in list1:
                add(wts, cidx - 1, -1.0 / len(list1))
            cen = x[[cc - 1 for cc in list1]].mean(axis=0)
            E = acc(E, _radial_term(x, wts, np.zeros(3), lim, e0, g, H), float(np.linalg.norm(x[a - 1] - cen)))
    else:
        raise ValueError(kind)
    return E, g, H
