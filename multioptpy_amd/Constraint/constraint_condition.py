"""ProjectOutConstrain: drop-in for ``multioptpy.Constraint.constraint_condition.ProjectOutConstrain`` (reference
constraint_condition.py:340-754): bond / fbond / angle / dihedral / x / y / z and the "experimental" kinds rot / eigvec / atoms_pair.

Host (this class, O(c n) scalar work as in the rest of the plugin layer): the constraint rows (the reference differentiates
torch_calc_distance / _angle / _dihedral_angle with autograd, Coordinate/redundant_coordinate.py:322-369; here the analytic Wilson
vectors -- equal up to the sign of a dihedral row, which no consumer sees), their orthonormal basis (numpy SVD of the c x 3N matrix, as
the reference), the gradient projection, the history of bases and the SHAKE-like coordinate adjustment.  Device: the O(n^2)
Hessian projection with its stiffness walls (``mopt_constraint_project_hess``; the reference forms three n^3 products).
Round 3: fbond (distance of two fragment centroids; the reference's atom list carries the split index as its last entry AND keeps it in
the second fragment as an atom label, :385-391 -- reproduced), rot (three rows per listed atom, :764-782, whose centred coordinates are
indexed with the GLOBAL atom index -- reproduced, so lists that are not 1..k raise IndexError or pick other atoms, as there), eigvec
(one eigenvector of the Hessian handed to initialize; the eigen-solve runs on the device, ops.eigh) and atoms_pair (:425-437).  The last
three keep the geometry OBJECT they were initialised with and align the current geometry with it by a Kabsch rotation before the
SHAKE-like loop (:481-497); that call centres both arrays in place, as Calculationtools.kabsch_algorithm does."""
from collections import deque

import numpy as np

from .. import _lib
from .._lib import lib, check, f64, ptr

_KINDS = ("bond", "fbond", "angle", "dihedral", "x", "y", "z", "rot", "eigvec", "atoms_pair")
_SPECIAL = ("rot", "eigvec", "atoms_pair")


def _angle(v1, v2):
    m1, m2 = np.linalg.norm(v1), np.linalg.norm(v2)
    m1 = m1 + 1e-15 if abs(m1) < 1e-15 else m1
    m2 = m2 + 1e-15 if abs(m2) < 1e-15 else m2
    return np.arccos(np.matmul(v1, v2) / (m1 * m2))


def _dihedral(v1, v2, v3):
    a, b = np.cross(v1, v2), np.cross(v2, v3)
    na, nb = np.linalg.norm(a), np.linalg.norm(b)
    na = na + 1e-15 if abs(na) < 1e-15 else na
    nb = nb + 1e-15 if abs(nb) < 1e-15 else nb
    ang = abs(np.arccos(np.sum(a * b) / (na * nb)))
    sign = np.sign(np.dot(np.cross(a / na, b / nb), v2))
    return -ang * sign if sign != 0 else ang


def _rotate(p, origin, axis, angle):
    u = axis / np.linalg.norm(axis)
    c, s = np.cos(angle), np.sin(angle)
    R = np.array([[c + u[0] ** 2 * (1 - c), u[0] * u[1] * (1 - c) - u[2] * s, u[0] * u[2] * (1 - c) + u[1] * s],
                  [u[1] * u[0] * (1 - c) + u[2] * s, c + u[1] ** 2 * (1 - c), u[1] * u[2] * (1 - c) - u[0] * s],
                  [u[2] * u[0] * (1 - c) - u[1] * s, u[2] * u[1] * (1 - c) + u[0] * s, c + u[2] ** 2 * (1 - c)]])
    return R @ (p - origin) + origin


def _unit(v):
    return v / np.linalg.norm(v)


class ProjectOutConstrain:
    def __init__(self, constraint_name, constraint_atoms_list, constraint_constant=[]):
        self.constraint_name = list(constraint_name)
        self.constraint_atoms_list = [[int(a) for a in lst] for lst in constraint_atoms_list]
        for nm in self.constraint_name:
            if nm not in _KINDS:
                raise NotImplementedError(f"constraint '{nm}' is not available in multioptpy_amd (available: {', '.join(_KINDS)})")
        self.constraint_constant = list(constraint_constant)
        self.iteration = 1
        self.init_tag = True
        self.spring_const = 0.0
        self.reference_scale = None
        self.alpha_smoothing = 0.7
        self.current_step = 0
        self.history_size = 5
        self.q_history = deque(maxlen=self.history_size)
        self.last_shake_iter = 0
        self.device = None
        self.projection_vec = None
        self.arbitrary_proj_vec = None

    @staticmethod
    def _fragments(at):
        """fbond: [a_1 .. a_k, split] -> 0-based fragments list[:split], list[split:] (the split index stays in the second one, :386-388)"""
        split = at[-1]
        return np.array(at[:split], dtype=np.int32) - 1, np.array(at[split:], dtype=np.int32) - 1

    # -- values of the constrained coordinates (constraint_condition.py:371-402) ----------------------------------------
    def _values(self, x, geom_obj=None, **kwargs):
        """Values of the constrained coordinates; the special kinds contribute the geometry object itself and (re)build their vectors."""
        out = []
        proj, arb = [], []
        for nm, at in zip(self.constraint_name, self.constraint_atoms_list):
            a = [i - 1 for i in at]
            if nm == "bond":
                out.append(np.linalg.norm(x[a[0]] - x[a[1]]))
            elif nm == "fbond":
                f1, f2 = self._fragments(at)
                out.append(np.linalg.norm(np.mean(x[f1], axis=0) - np.mean(x[f2], axis=0)))
            elif nm == "angle":
                out.append(_angle(x[a[0]] - x[a[1]], x[a[2]] - x[a[1]]))
            elif nm == "dihedral":
                out.append(_dihedral(x[a[0]] - x[a[1]], x[a[1]] - x[a[2]], x[a[2]] - x[a[3]]))
            elif nm in ("x", "y", "z"):
                out.append(x[a[0]]["xyz".index(nm)])
            elif nm == "rot":
                out.append(geom_obj)
            elif nm == "eigvec":
                if kwargs.get("hessian", None) is None:
                    raise Exception("error (Hessian is required for eigvec constraint)")
                from .. import ops
                w, v = ops.eigh(np.array(kwargs["hessian"], dtype=np.float64))
                valid = np.where(np.abs(w) > 1.0e-10)[0]
                order = valid[np.argsort(w[valid])]
                proj.append(np.array(v[:, order[int(at[0])]]))
                out.append(geom_obj)
            else:                                                    # atoms_pair
                vec = np.zeros_like(x)
                vec[a[0]] = -x[a[0]] + x[a[1]]
                vec[a[1]] = -x[a[1]] + x[a[0]]
                nrm = np.linalg.norm(vec)
                if nrm < 1.0e-10:
                    raise Exception("error (the distance between the pair atoms is too small)")
                arb.append((vec / nrm).reshape(-1, 1))
                out.append(geom_obj)
        self.projection_vec = proj
        ortho = []                                                   # Gram-Schmidt of the pair vectors (:443-454)
        for v in arb:
            w_ = v.copy()
            for u in ortho:
                w_ -= np.dot(u.T, w_) * u
            nrm = np.linalg.norm(w_)
            if nrm > 1e-10:
                ortho.append(w_ / nrm)
        self.arbitrary_proj_vec = ortho
        return out

    def initialize(self, geom_num_list, **kwargs):
        vals = self._values(np.asarray(geom_num_list, dtype=np.float64), geom_obj=geom_num_list, **kwargs)
        if self.init_tag:
            if len(self.constraint_constant) == 0:
                self.init_constraint = list(vals)
            else:
                self.init_constraint = []
                for nm, cst in zip(self.constraint_name, self.constraint_constant):
                    if nm in ("bond", "fbond", "x", "y", "z"):
                        self.init_constraint.append(cst / 0.52917721067)
                    elif nm in ("angle", "dihedral"):
                        self.init_constraint.append(np.deg2rad(cst))
                    else:
                        self.init_constraint.append(geom_num_list)
            self.init_tag = False
        return vals

    # -- SHAKE-like adjustment (constraint_condition.py:473-541), in place like the reference ---------------------------
    def adjust_init_coord(self, coord, hessian=None):
        if self.init_tag or not hasattr(self, "init_constraint"):
            self.initialize(coord, hessian=hessian)
        x = coord
        # the special kinds first: align with the geometry they were initialised with (:481-497)
        for nm, at, tgt in zip(self.constraint_name, self.constraint_atoms_list, self.init_constraint):
            if nm == "rot":
                from ..Utils.calc_tools import Calculationtools
                t = np.array(at, dtype=np.int32) - 1
                centre = np.mean(x[t], axis=0)
                part, ref = x[t], np.asarray(tgt)[t]                  # (fancy indexing: copies, as in rotate_partial_struct :788-797)
                rot, _ = Calculationtools().kabsch_algorithm(part, ref)
                x[t] = rot + centre
            elif nm in ("eigvec", "atoms_pair"):
                from ..Utils.calc_tools import Calculationtools
                x, _ = Calculationtools().kabsch_algorithm(x, tgt)   # (centres x and the stored geometry in place, returns the rotated copy)
        plain = [i for i, nm in enumerate(self.constraint_name) if nm not in _SPECIAL]
        final = 0
        for jter in range(10000):
            final = jter
            for nm, at, tgt in zip(self.constraint_name, self.constraint_atoms_list, self.init_constraint):
                a = [i - 1 for i in at]
                if nm == "bond":
                    v = x[a[1]] - x[a[0]]
                    r = np.linalg.norm(v)
                    u = v / r
                    x[a[0]] = x[a[0]] - (tgt - r) * u * 0.5
                    x[a[1]] = x[a[1]] + (tgt - r) * u * 0.5
                elif nm == "fbond":                                  # change_fragm_distance_both_side, calc_tools.py:760-782
                    f1, f2 = self._fragments(at)
                    c1 = np.zeros(3)
                    for i in f1:
                        c1 += x[i]
                    c1 /= len(f1)
                    c2 = np.zeros(3)
                    for i in f2:
                        c2 += x[i]
                    c2 /= len(f2)
                    v = c2 - c1
                    r = np.linalg.norm(v)
                    u = v / r
                    for i in f1:
                        x[i] = x[i] - (tgt - r) * u * 0.5
                    for i in f2:
                        x[i] = x[i] + (tgt - r) * u * 0.5
                elif nm == "angle":
                    BA, BC = x[a[0]] - x[a[1]], x[a[2]] - x[a[1]]
                    diff = tgt - _angle(BA, BC)
                    axis = _unit(np.cross(BA, BC))
                    nA, nC = _rotate(x[a[0]], x[a[1]], axis, -diff / 2.0), _rotate(x[a[2]], x[a[1]], axis, diff / 2.0)
                    x[a[0]], x[a[2]] = nA, nC
                elif nm == "dihedral":
                    diff = tgt - _dihedral(x[a[0]] - x[a[1]], x[a[1]] - x[a[2]], x[a[2]] - x[a[3]])
                    BC = x[a[2]] - x[a[1]]
                    nD, nA = _rotate(x[a[3]], x[a[2]], BC, diff * 0.5), _rotate(x[a[0]], x[a[1]], BC, -diff * 0.5)
                    x[a[3]], x[a[0]] = nD, nA
                elif nm in ("x", "y", "z"):
                    x[a[0]]["xyz".index(nm)] = tgt
            vals = self.initialize(x, hessian=hessian)
            if not plain:
                break
            cur = np.array([vals[i] for i in plain], dtype=float)
            if np.linalg.norm(cur - np.array([self.init_constraint[i] for i in plain], dtype=float)) < 1.0e-10:
                break
        self.last_shake_iter = final
        return x

    # -- constraint rows and their basis (constraint_condition.py:543-637) ----------------------------------------------
    def _get_all_constraint_vectors(self, coord):
        x = np.asarray(coord, dtype=np.float64).reshape(-1, 3)
        n = x.size
        rows = []
        n_proj = n_arb = 0
        for nm, at in zip(self.constraint_name, self.constraint_atoms_list):
            a = [i - 1 for i in at]
            v = np.zeros(n)
            if nm == "bond":
                u = _unit(x[a[0]] - x[a[1]])
                v[3 * a[0]:3 * a[0] + 3] = u
                v[3 * a[1]:3 * a[1] + 3] = -u
            elif nm == "fbond":
                # d |c_1 - c_2| / dx_a = u (multiplicity of a in fragment 1 / |fragment 1| - the same for fragment 2)
                f1, f2 = self._fragments(at)
                u = _unit(np.mean(x[f1], axis=0) - np.mean(x[f2], axis=0))
                for i in f1:
                    v[3 * i:3 * i + 3] += u / len(f1)
                for i in f2:
                    v[3 * i:3 * i + 3] -= u / len(f2)
            elif nm == "rot":
                # constract_partial_rot_B_mat (:764-782): three rows per listed atom; the centred coordinates are indexed with the GLOBAL index
                t = np.array(at, dtype=np.int32) - 1
                cg = x[t] - np.mean(x[t], axis=0)
                for i in t:
                    r0, r1, r2 = np.zeros(n), np.zeros(n), np.zeros(n)
                    r0[3 * i + 1], r0[3 * i + 2] = cg[i][2], -cg[i][1]
                    r1[3 * i + 0], r1[3 * i + 2] = -cg[i][2], cg[i][0]
                    r2[3 * i + 0], r2[3 * i + 1] = cg[i][1], -cg[i][0]
                    rows.extend([r0, r1, r2])
                continue
            elif nm == "eigvec":
                if self.projection_vec is not None and n_proj < len(self.projection_vec):
                    rows.append(np.asarray(self.projection_vec[n_proj]))
                    n_proj += 1
                continue
            elif nm == "atoms_pair":
                if self.arbitrary_proj_vec is not None and n_arb < len(self.arbitrary_proj_vec):
                    rows.append(self.arbitrary_proj_vec[n_arb].reshape(-1))
                    n_arb += 1
                continue
            elif nm == "angle":
                e1, e2 = x[a[0]] - x[a[1]], x[a[2]] - x[a[1]]
                r1, r2 = np.linalg.norm(e1), np.linalg.norm(e2)
                e1, e2 = e1 / r1, e2 / r2
                co = float(np.dot(e1, e2))
                si = np.sqrt(1.0 - co * co)
                b0, b2 = (co * e1 - e2) / (r1 * si), (co * e2 - e1) / (r2 * si)
                v[3 * a[0]:3 * a[0] + 3] += b0
                v[3 * a[2]:3 * a[2] + 3] += b2
                v[3 * a[1]:3 * a[1] + 3] += -(b0 + b2)
            elif nm == "dihedral":
                # Wilson torsion vectors from the unit bond vectors and the two bond angles
                xi, xj, xk, xl = x[a[0]], x[a[1]], x[a[2]], x[a[3]]
                r1, r2, r3 = np.linalg.norm(xi - xj), np.linalg.norm(xj - xk), np.linalg.norm(xk - xl)
                uij, ujk, ukl = (xi - xj) / r1, (xj - xk) / r2, (xk - xl) / r3
                c2, c3 = float(np.dot(-uij, ujk)), float(np.dot(-ujk, ukl))          # cosines of the angles i-j-k and j-k-l
                s2sq, s3sq = 1.0 - c2 * c2, 1.0 - c3 * c3
                t0 = np.cross(uij, ujk) / (r1 * s2sq)
                t3 = np.cross(-ukl, -ujk) / (r3 * s3sq)
                t1 = -((r2 - r1 * c2) * t0 + r3 * c3 * t3) / r2
                t2 = -(t0 + t1 + t3)
                for k, dv in zip(a, (t0, t1, t2, t3)):
                    v[3 * k:3 * k + 3] += dv
            else:
                v[3 * a[0] + "xyz".index(nm)] = 1.0
            rows.append(v)
        return np.array(rows) if rows else None

    def _get_orthonormal_basis(self, coord):
        B = self._get_all_constraint_vectors(coord)
        if B is None:
            return None
        try:
            U, S, _ = np.linalg.svd(B.T, full_matrices=False)
        except np.linalg.LinAlgError:
            return None
        rank = int(np.sum(S > 1e-6))
        return U[:, :rank] if rank > 0 else None

    def calc_project_out_grad(self, coord, grad):
        natom = len(coord)
        g = np.asarray(grad, dtype=np.float64).reshape(3 * natom, 1)
        Q = self._get_orthonormal_basis(coord)
        if Q is None:
            return grad
        g = g - Q @ (Q.T @ g)
        g = g - Q @ (Q.T @ g)           # purification
        return g.reshape(natom, 3)

    def calc_project_out_hess(self, coord, grad, hessian):
        Q = self._get_orthonormal_basis(coord)
        if Q is None:
            return hessian
        self.q_history.append(Q)
        vecs = [Q[:, i] for i in range(Q.shape[1])]
        for Qh in self.q_history:
            for i in range(Qh.shape[1]):
                v = Qh[:, i].copy()
                for b in vecs:
                    v = v - np.dot(v, b) * b
                nrm = np.linalg.norm(v)
                if nrm > 0.1:
                    vecs.append(v / nrm)
        nh = Q.shape[1]
        self.current_step += 1
        H = np.asarray(hessian, dtype=np.float64)
        scale = max(float(np.max(np.abs(np.diag(H)))), 0.5)
        self.reference_scale = scale if self.reference_scale is None else (self.alpha_smoothing * self.reference_scale + (1.0 - self.alpha_smoothing) * scale)
        n = H.shape[0]
        Qh = f64(np.array(vecs[:nh]))
        Qs = f64(np.array(vecs[nh:])) if len(vecs) > nh else None
        out = np.empty((n, n))
        c = _lib.ctx(self.device)
        check(lib().mopt_constraint_project_hess(c, n, ptr(f64(H)), nh, ptr(Qh), 0 if Qs is None else Qs.shape[0], ptr(Qs),
                                                 self.reference_scale * 100.0, self.reference_scale * 100.0 * 0.2, ptr(out)), c)
        return out

    def reset_stiffness(self):
        self.reference_scale = None
        self.current_step = 0
        self.q_history.clear()
        self.last_shake_iter = 0
