"""ORACLE (test infrastructure, never imported by the product): NumPy restatement of the reference's constraint projection and of the
constrained RS-I-RFO optimizer
  ProjectOutConstrain   multioptpy/Constraint/constraint_condition.py:340-754   (bond / angle / dihedral / x / y / z constraints)
  CRSIRFO               multioptpy/Optimizer/crsirfo.py:15-182
with the helpers they use (Utils/calc_tools.py:514-625, 750-758; Coordinate/redundant_coordinate.py:322-369: the reference takes the
constraint gradients by torch autograd, here they are the analytic Wilson vectors -- equal up to the sign of a dihedral row, which no
consumer sees: only the span of the rows enters).  Pinned by tests/golden/g12_constraints.npz (outputs of the reference)."""
from collections import deque

import numpy as np
import scipy.linalg

from .model_hessian import stretch2, bend2, torsion2
from .rsirfo import RSIRFO


# ---- internal-coordinate values and the SHAKE-like moves (Utils/calc_tools.py) ---------------------------------------------------
def angle_from_vec(v1, v2):
    m1, m2 = np.linalg.norm(v1), np.linalg.norm(v2)
    m1 = m1 + 1e-15 if abs(m1) < 1e-15 else m1
    m2 = m2 + 1e-15 if abs(m2) < 1e-15 else m2
    return np.arccos(np.matmul(v1, v2) / (m1 * m2))


def dihedral_from_vec(v1, v2, v3):
    a, b = np.cross(v1, v2), np.cross(v2, v3)
    na, nb = np.linalg.norm(a), np.linalg.norm(b)
    na = na + 1e-15 if abs(na) < 1e-15 else na
    nb = nb + 1e-15 if abs(nb) < 1e-15 else nb
    ang = abs(np.arccos(np.sum(a * b) / (na * nb)))
    sign = np.sign(np.dot(np.cross(a / na, b / nb), v2))
    return -ang * sign if sign != 0 else ang


def rotate_atom(p, origin, axis, angle):
    u = axis / np.linalg.norm(axis)
    c, s = np.cos(angle), np.sin(angle)
    R = np.array([[c + u[0] ** 2 * (1 - c), u[0] * u[1] * (1 - c) - u[2] * s, u[0] * u[2] * (1 - c) + u[1] * s],
                  [u[1] * u[0] * (1 - c) + u[2] * s, c + u[1] ** 2 * (1 - c), u[1] * u[2] * (1 - c) - u[0] * s],
                  [u[2] * u[0] * (1 - c) - u[1] * s, u[2] * u[1] * (1 - c) + u[0] * s, c + u[2] ** 2 * (1 - c)]])
    return R @ (p - origin) + origin


def set_distance(x, a, b, target):
    v = x[b] - x[a]
    r = np.linalg.norm(v)
    u = v / r
    x[a] = x[a] - (target - r) * u * 0.5
    x[b] = x[b] + (target - r) * u * 0.5
    return x


def set_angle(x, a, b, c, target):
    BA, BC = x[a] - x[b], x[c] - x[b]
    cur = angle_from_vec(BA, BC)
    axis = np.cross(BA, BC)
    axis = axis / np.linalg.norm(axis)
    diff = target - cur
    na, nc = rotate_atom(x[a], x[b], axis, -diff / 2.0), rotate_atom(x[c], x[b], axis, diff / 2.0)
    x[a], x[c] = na, nc
    return x


def set_dihedral(x, a, b, c, d, target):
    cur = dihedral_from_vec(x[a] - x[b], x[b] - x[c], x[c] - x[d])
    diff = target - cur
    BC = x[c] - x[b]
    nd, na = rotate_atom(x[d], x[c], BC, diff * 0.5), rotate_atom(x[a], x[b], BC, -diff * 0.5)
    x[d], x[a] = nd, na
    return x


class ProjectOutConstrain:
    def __init__(self, constraint_name, constraint_atoms_list, constraint_constant=()):
        self.names = list(constraint_name)
        self.atoms = [[int(a) for a in lst] for lst in constraint_atoms_list]
        for nm in self.names:
            if nm not in ("bond", "fbond", "angle", "dihedral", "x", "y", "z", "rot", "eigvec", "atoms_pair"):
                raise NotImplementedError(f"constraint '{nm}' is not restated by the oracle")
        self.projection_vec = []
        self.arbitrary_proj_vec = []
        self.init_tag = True
        self.reference_scale = None
        self.alpha_smoothing = 0.7
        self.current_step = 0
        self.q_history = deque(maxlen=5)
        self.last_shake_iter = 0
        self.init_constraint = None

    @staticmethod
    def kabsch(P, Q):
        """Utils/calc_tools.py:412-425: both arguments are centred IN PLACE; returns (rotated P, Q)."""
        P -= np.array([np.mean(P.T[0]), np.mean(P.T[1]), np.mean(P.T[2])])
        Q -= np.array([np.mean(Q.T[0]), np.mean(Q.T[1]), np.mean(Q.T[2])])
        U, S, Vt = np.linalg.svd(np.dot(P.T, Q))
        R = np.dot(Vt.T, U.T)
        if np.linalg.det(R) < 0:
            Vt[-1, :] *= -1
            R = np.dot(Vt.T, U.T)
        return np.dot(R, P.T).T, Q

    def values(self, x, obj=None, hessian=None):
        """constraint_condition.py:371-441 (obj: the geometry object the special kinds keep)"""
        out = []
        self.projection_vec = []
        arb = []
        for nm, at in zip(self.names, self.atoms):
            a = [i - 1 for i in at]
            if nm == "bond":
                out.append(np.linalg.norm(x[a[0]] - x[a[1]]))
            elif nm == "fbond":
                d = at[-1]
                f1, f2 = np.array(at[:d], dtype=np.int32) - 1, np.array(at[d:], dtype=np.int32) - 1
                out.append(np.linalg.norm(np.mean(x[f1], axis=0) - np.mean(x[f2], axis=0)))
            elif nm == "angle":
                out.append(angle_from_vec(x[a[0]] - x[a[1]], x[a[2]] - x[a[1]]))
            elif nm == "dihedral":
                out.append(dihedral_from_vec(x[a[0]] - x[a[1]], x[a[1]] - x[a[2]], x[a[2]] - x[a[3]]))
            elif nm in ("x", "y", "z"):
                out.append(x[a[0]]["xyz".index(nm)])
            elif nm == "rot":
                out.append(obj)
            elif nm == "eigvec":
                if hessian is None:
                    raise Exception("error (Hessian is required for eigvec constraint)")
                w, v = np.linalg.eigh(np.array(hessian))
                valid = np.where(np.abs(w) > 1.0e-10)[0]
                srt = valid[np.argsort(w[valid])]
                self.projection_vec.append(v[:, srt[int(at[0])]])
                out.append(obj)
            else:
                vec = np.zeros_like(x)
                vec[a[0]] = -x[a[0]] + x[a[1]]
                vec[a[1]] = -x[a[1]] + x[a[0]]
                nv = np.linalg.norm(vec)
                if nv < 1.0e-10:
                    raise Exception("error (the distance between the pair atoms is too small)")
                arb.append((vec / nv).reshape(-1, 1))
                out.append(obj)
        ortho = []
        for v in arb:
            w_ = v.copy()
            for u in ortho:
                w_ -= np.dot(u.T, w_) * u
            nw = np.linalg.norm(w_)
            if nw > 1e-10:
                ortho.append(w_ / nw)
        self.arbitrary_proj_vec = ortho
        return out

    def initialize(self, x, **kw):
        vals = self.values(np.asarray(x, dtype=float), obj=x, hessian=kw.get("hessian"))
        if self.init_tag:
            self.init_constraint = list(vals)
            self.init_tag = False
        return vals

    def adjust_init_coord(self, coord, hessian=None):
        """Works IN PLACE on a float64 ndarray, as the reference does (its change_*_both_side helpers assign into the array they are
        given, Utils/calc_tools.py:750-758) -- CRSIRFO relies on that, see there."""
        x = coord if isinstance(coord, np.ndarray) and coord.dtype == np.float64 else np.array(coord, dtype=float)
        if self.init_tag or self.init_constraint is None:
            self.initialize(x, hessian=hessian)
        for nm, at, tgt in zip(self.names, self.atoms, self.init_constraint):          # constraint_condition.py:481-497
            if nm == "rot":
                t = np.array(at, dtype=np.int32) - 1
                centre = np.mean(x[t], axis=0)
                rot, _ = self.kabsch(x[t], tgt[t])
                x[t] = rot + centre
            elif nm in ("eigvec", "atoms_pair"):
                x, _ = self.kabsch(x, tgt)
        special = ("rot", "eigvec", "atoms_pair")
        final = 0
        for jter in range(10000):
            final = jter
            for nm, at, tgt in zip(self.names, self.atoms, self.init_constraint):
                a = [i - 1 for i in at]
                if nm == "bond":
                    x = set_distance(x, a[0], a[1], tgt)
                elif nm == "fbond":
                    d = at[-1]
                    f1, f2 = np.array(at[:d], dtype=np.int32) - 1, np.array(at[d:], dtype=np.int32) - 1
                    c1 = sum((x[i] for i in f1), np.zeros(3)) / len(f1)
                    c2 = sum((x[i] for i in f2), np.zeros(3)) / len(f2)
                    vec = c2 - c1
                    nv = np.linalg.norm(vec)
                    for i in f1:
                        x[i] = x[i] - (tgt - nv) * (vec / nv) * 0.5
                    for i in f2:
                        x[i] = x[i] + (tgt - nv) * (vec / nv) * 0.5
                elif nm == "angle":
                    x = set_angle(x, a[0], a[1], a[2], tgt)
                elif nm == "dihedral":
                    x = set_dihedral(x, a[0], a[1], a[2], a[3], tgt)
                elif nm in ("x", "y", "z"):
                    x[a[0]]["xyz".index(nm)] = tgt
            vals = self.initialize(x, hessian=hessian)
            cur = [v for v, nm in zip(vals, self.names) if nm not in special]
            ref = [v for v, nm in zip(self.init_constraint, self.names) if nm not in special]
            if len(cur) == 0 or np.linalg.norm(np.array(cur, dtype=float) - np.array(ref, dtype=float)) < 1.0e-10:
                break
        self.last_shake_iter = final
        return x

    def constraint_vectors(self, x):
        x = np.asarray(x, dtype=float).reshape(-1, 3)
        n = x.size
        rows = []
        npv = nav = 0
        for nm, at in zip(self.names, self.atoms):
            a = [i - 1 for i in at]
            v = np.zeros(n)
            if nm == "fbond":
                # the reference differentiates |mean(x[f1]) - mean(x[f2])| with autograd (redundant_coordinate.py:328-332)
                d = at[-1]
                f1, f2 = np.array(at[:d], dtype=np.int32) - 1, np.array(at[d:], dtype=np.int32) - 1
                dv = np.mean(x[f1], axis=0) - np.mean(x[f2], axis=0)
                u = dv / np.linalg.norm(dv)
                for i in f1:
                    v[3 * i:3 * i + 3] += u / len(f1)
                for i in f2:
                    v[3 * i:3 * i + 3] -= u / len(f2)
                rows.append(v)
                continue
            if nm == "rot":                                        # constraint_condition.py:764-782
                t = np.array(at, dtype=np.int32) - 1
                cg = x[t] - np.mean(x[t], axis=0)
                Bm = np.zeros((3 * len(t), n))
                for j in range(len(t)):
                    i = t[j]
                    Bm[3 * j][3 * i + 1] = cg[i][2]
                    Bm[3 * j][3 * i + 2] = -1 * cg[i][1]
                    Bm[3 * j + 1][3 * i + 0] = -1 * cg[i][2]
                    Bm[3 * j + 1][3 * i + 2] = cg[i][0]
                    Bm[3 * j + 2][3 * i + 0] = cg[i][1]
                    Bm[3 * j + 2][3 * i + 1] = -1 * cg[i][0]
                rows.extend(list(Bm))
                continue
            if nm == "eigvec":
                if npv < len(self.projection_vec):
                    rows.append(self.projection_vec[npv])
                    npv += 1
                continue
            if nm == "atoms_pair":
                if nav < len(self.arbitrary_proj_vec):
                    rows.append(self.arbitrary_proj_vec[nav].reshape(-1))
                    nav += 1
                continue
            if nm == "bond":
                _, b = stretch2(x[a])
                for k, i in enumerate(a):
                    v[3 * i:3 * i + 3] = b[k] * -1.0          # d|x_1 - x_2| / dx_1 = (x_1 - x_2) / r
            elif nm == "angle":
                _, b = bend2(x[a])
                for k, i in enumerate(a):
                    v[3 * i:3 * i + 3] += b[k]
            elif nm == "dihedral":
                b = torsion2(x[a])
                for k, i in enumerate(a):
                    v[3 * i:3 * i + 3] += b[k]
            else:
                v[3 * a[0] + "xyz".index(nm)] = 1.0
            rows.append(v)
        return np.array(rows) if rows else None

    def orthonormal_basis(self, x):
        B = self.constraint_vectors(x)
        if B is None:
            return None
        U, S, _ = np.linalg.svd(B.T, full_matrices=False)
        rank = int(np.sum(S > 1e-6))
        return U[:, :rank] if rank > 0 else None

    def calc_project_out_grad(self, coord, grad):
        N = len(coord)
        g = np.asarray(grad, dtype=float).reshape(3 * N, 1)
        Q = self.orthonormal_basis(coord)
        if Q is None:
            return grad
        g = g - Q @ (Q.T @ g)
        g = g - Q @ (Q.T @ g)
        return g.reshape(N, 3)

    def exclusion_vectors(self, Q):
        """Current basis (hard wall) + the parts of the stored bases orthogonal to everything collected so far with norm > 0.1 (soft wall)."""
        vecs = [Q[:, i] for i in range(Q.shape[1])]
        for Qh in self.q_history:
            for i in range(Qh.shape[1]):
                v = Qh[:, i].copy()
                for b in vecs:
                    v = v - np.dot(v, b) * b
                nrm = np.linalg.norm(v)
                if nrm > 0.1:
                    vecs.append(v / nrm)
        return vecs

    def calc_project_out_hess(self, coord, grad, hessian):
        Q = self.orthonormal_basis(coord)
        if Q is None:
            return hessian
        self.q_history.append(Q)
        vecs = self.exclusion_vectors(Q)
        nh = Q.shape[1]
        Ph = sum(np.outer(v, v) for v in vecs[:nh])
        Ps = sum((np.outer(v, v) for v in vecs[nh:]), np.zeros_like(hessian))
        PHP = hessian - Ph @ hessian - hessian @ Ph + Ph @ (hessian @ Ph)
        self.current_step += 1
        scale = max(np.max(np.abs(np.diag(hessian))), 0.5)
        self.reference_scale = scale if self.reference_scale is None else self.alpha_smoothing * self.reference_scale + (1.0 - self.alpha_smoothing) * scale
        return PHP + self.reference_scale * 100.0 * Ph + self.reference_scale * 20.0 * Ps


class CRSIRFO(RSIRFO):
    """crsirfo.py:15-182: SHAKE-like correction with gradient transport, Hessian update, RFO in the null space of the constraints."""

    def __init__(self, constraints=None, **config):
        super().__init__(**config)
        self.constraints_obj = constraints
        self.svd_threshold = config.get("svd_threshold", 1e-5)
        self.proj_grad_converged = False

    def null_space(self, x):
        n = x.size
        if self.constraints_obj is None:
            return np.eye(n)
        B = self.constraints_obj.constraint_vectors(x.reshape(-1, 3))
        if B is None or len(B) == 0:
            return np.eye(n)
        nrm = np.linalg.norm(B, axis=1)
        nrm[nrm < 1e-12] = 1.0
        Bn = B / nrm[:, None]
        U, S, _ = scipy.linalg.svd(Bn.T, full_matrices=True)
        thr = max(self.svd_threshold, (S[0] if len(S) else 1.0) * 1e-6)
        return U[:, int(np.sum(S > thr)):]

    def run(self, geom_num_list, B_g, pre_B_g=[], pre_geom=[], B_e=0.0, pre_B_e=0.0, pre_move_vector=[], initial_geom_num_list=[], g=[], pre_g=[]):
        if self.Initialization:
            self.prev_eigvec_size = None
            self.predicted_energy_changes = []
            self.actual_energy_changes = []
            self.prev_geometry = self.prev_gradient = self.prev_energy = None
            self.proj_grad_converged = False
            self.iteration = 0
            self.Initialization = False
        grad = np.asarray(B_g, dtype=float).ravel()
        shape = np.asarray(geom_num_list).shape
        flat = geom_num_list.ravel()                     # a VIEW of the caller's array (crsirfo.py:62) ...
        if self.constraints_obj is not None:
            # ... and so is the (N,3) array handed to the SHAKE-like adjustment, which assigns into it: the caller's geometry is
            # corrected in place and the "displacement" below is corrected - corrected = 0 for every contiguous ndarray input, so the
            # gradient transport of crsirfo.py:71-83 never fires.  Reproduced as is.
            corrected = self.constraints_obj.adjust_init_coord(geom_num_list.reshape(-1, 3)).ravel()
            disp = corrected - flat
            if np.linalg.norm(disp) > 1e-6:
                H_eff = self.hessian
                if self.bias_hessian is not None:
                    H_eff += self.bias_hessian           # (in place, as in the reference: crsirfo.py:79-81)
                grad = grad + H_eff @ disp
            geom_num_list = corrected.reshape(shape)
        if self.prev_geometry is not None and self.prev_gradient is not None and len(pre_g) > 0 and len(pre_geom) > 0:
            self.update_hessian(geom_num_list, g, pre_geom, pre_g)
        H = self.hessian
        if self.bias_hessian is not None:
            H += self.bias_hessian
        U = self.null_space(np.asarray(geom_num_list, dtype=float).ravel())
        if U.shape[1] == 0:
            return np.zeros_like(grad).reshape(-1, 1)
        gs = U.T @ grad
        Hs = U.T @ (H @ U)
        gnorm = np.linalg.norm(gs)
        if gnorm < self.gradient_norm_threshold:
            self.proj_grad_converged = True
            self.prev_geometry, self.prev_gradient, self.prev_energy = geom_num_list, B_g, B_e
            return np.zeros_like(grad).reshape(-1, 1)
        Hs = 0.5 * (Hs + Hs.T)
        w, v = self.eigh_with_shift(Hs)
        if self.prev_energy is not None:
            actual = B_e - self.prev_energy
            if len(self.actual_energy_changes) >= 3:
                self.actual_energy_changes.pop(0)
            self.actual_energy_changes.append(actual)
            if self.predicted_energy_changes:
                self.adjust_trust_radius(actual, self.predicted_energy_changes[-1], w[0] if len(w) > 0 else None, gnorm)
        m = len(gs)
        P = np.eye(m)
        k = i = 0
        while k < len(self.roots) and i < len(w):
            if abs(w[i]) > 1e-10:
                P -= (1.0 if self.NEB_mode else 2.0) * np.outer(v[:, i], v[:, i])
                k += 1
            i += 1
        Hstar = P @ Hs
        Hstar = 0.5 * (Hstar + Hstar.T)
        gstar = P @ gs
        ws, vs = self.eigh_with_shift(Hstar)
        keep = ~(np.abs(ws) < self.small_eigval_thresh)
        ws, vs = ws[keep], vs[:, keep]
        step_sub = self.get_rs_step(ws, vs, gstar)
        step = U @ step_sub
        pred = gs @ step_sub + 0.5 * (step_sub @ Hs @ step_sub)
        if len(self.predicted_energy_changes) >= 3:
            self.predicted_energy_changes.pop(0)
        self.predicted_energy_changes.append(pred)
        self.prev_geometry, self.prev_gradient, self.prev_energy = geom_num_list, B_g, B_e
        self.iteration += 1
        return -1 * step.reshape(-1, 1)
